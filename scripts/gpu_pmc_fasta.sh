#!/bin/bash
# Two PMC passes (SQ instruction mix, waits) of the FASTA bench; per-kernel averages.  Usage: scripts/gpu_pmc_fasta.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcfa
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
B="python bench.py --fasta --no-cpu-baseline --steps 2 --warmup 1 --min-seconds 0"
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -d $OUT/a -o q -- $B) > $OUT/a.log 2>&1 </dev/null
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS -d $OUT/b -o q -- $B) > $OUT/b.log 2>&1 </dev/null
for d in a b; do timeout 60 python scripts/summarize_pmc.py $OUT/$d k_fa_emit </dev/null; timeout 60 python scripts/summarize_pmc.py $OUT/$d k_fa_tile_sums </dev/null; done
