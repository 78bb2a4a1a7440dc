#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of the bench command
# (PMC never combined with other tracing, per the pool rules).  Usage: scripts/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
BENCH="python bench.py --no-cpu-baseline --no-extra-modes $*"   # (the extra figures of the default line run other inputs through the same kernels: they would blur the per-kernel averages)
LONG="--steps 10 --warmup 3"; SHORT="--steps 2 --warmup 1 --min-seconds 0"   # PMC passes: no time-based warm-up (the counter databases grow with every dispatch)
# another command under the same passes (e.g. scripts/bench_fasta.py): BZQ_PROFILE_CMD="python scripts/bench_fasta.py 0"
if [ -n "${BZQ_PROFILE_CMD:-}" ]; then BENCH="$BZQ_PROFILE_CMD"; LONG=""; SHORT=""; fi
(timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o $TAG -- $BENCH $LONG) > $OUT/kt.log 2>&1; echo "kt rc=$?"
(timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq -o $TAG -- $BENCH $SHORT) > $OUT/pmc_sq.log 2>&1; echo "pmc_sq rc=$?"
(timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM -d $OUT/pmc_sq2 -o $TAG -- $BENCH $SHORT) > $OUT/pmc_sq2.log 2>&1; echo "pmc_sq2 rc=$?"
(timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o $TAG -- $BENCH $SHORT) > $OUT/pmc_fetch.log 2>&1; echo "pmc_fetch rc=$?"
(timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o $TAG -- $BENCH $SHORT) > $OUT/pmc_write.log 2>&1; echo "pmc_write rc=$?"
grep "^{\"metric\"" $OUT/kt.log | tail -1 > $OUT/bench_line_under_profiler.json
find $OUT -type f -size +8M -delete
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
