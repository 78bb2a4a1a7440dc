#!/bin/bash
# HBM traffic of the FASTA kernels (FETCH_SIZE / WRITE_SIZE in separate passes).  Usage: scripts/gpu_traffic_fasta.sh [lib.so]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trfa
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
[ -n "${1:-}" ] && export BLAZESEQ_HIP_LIB=$R/$1
B="python bench.py --fasta --no-cpu-baseline --steps 2 --warmup 1 --min-seconds 0"
(timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o q -- $B) > $OUT/f.log 2>&1 </dev/null
(timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -o q -- $B) > $OUT/w.log 2>&1 </dev/null
for d in f w; do timeout 60 python scripts/summarize_pmc.py $OUT/$d k_fa_emit </dev/null; timeout 60 python scripts/summarize_pmc.py $OUT/$d k_fa_tile_sums </dev/null; done
