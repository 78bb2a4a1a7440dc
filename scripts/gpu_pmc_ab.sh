#!/bin/bash
# A/B of PMC counters for bench variants. Usage: gpu_pmc_ab.sh "<bench args A>" "<bench args B>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp; cd $R
i=0
for ARGS in "$@"; do
  i=$((i+1))
  OUT=$R/gpurun_out/ab_$i; rm -rf $OUT; mkdir -p $OUT
  (timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/a -o x -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 $ARGS) > $OUT/a.log 2>&1
  (timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/b -o x -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 $ARGS) > $OUT/b.log 2>&1
  (timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL -d $OUT/c -o x -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 $ARGS) > $OUT/c.log 2>&1
  echo "=== variant $i: $ARGS"
  python - <<PY
import sqlite3,glob
for sub in 'abc':
    for f in glob.glob('$OUT/'+sub+'/*.db'):
        db=sqlite3.connect(f)
        for k,c,v,n in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%k_fused%' or kernel_name like '%aggregate2%' group by kernel_name, counter_name"):
            print("  %-28s %-30s %16.0f" % (k.split('(')[0][-28:], c, v))
PY
done
