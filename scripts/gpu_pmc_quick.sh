#!/bin/bash
# One PMC pass (SQ instruction mix) of the bench command; prints per-kernel averages.  Usage: scripts/gpu_pmc_quick.sh [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -o q -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 $*) > $OUT/pmc.log 2>&1
python scripts/summarize_prof.py $OUT 2>&1 | grep -A9 "k_fused\|k_tile_aggregate2\|k_views<\|k_tile_count\|k_tile_lines\|k_views_join" | grep -v "^--"
