#!/bin/bash
# Write-path counters of the FASTQ emit and the FASTA emit side by side (a counter-backed reason for the FASTA
# emit's 4.2 TB/s against the FASTQ emit's 5.7).  Separate --pmc passes, kernel trace only (pool rules).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcw
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TCC|TA|TD)_[A-Z0-9_]+" | sort -u > $OUT/counters_available.txt
FQ="python bench.py --no-cpu-baseline --no-extra-modes --steps 2 --warmup 1 --min-seconds 0"
FA="python bench.py --fasta --no-cpu-baseline --steps 2 --warmup 1 --min-seconds 0"
i=0
for SET in "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_WRITEBACK_sum" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  (timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/fq$i -o q -- $FQ) > $OUT/fq$i.log 2>&1
  (timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/fa$i -o q -- $FA) > $OUT/fa$i.log 2>&1
done
find $OUT -type f -size +8M -delete
for d in $OUT/fq* $OUT/fa*; do [ -d $d ] && { echo "== $(basename $d)"; python scripts/summarize_pmc.py $d k_f | grep -E -A8 "k_fused|k_fa_emit" | grep -v "^--"; }; done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
