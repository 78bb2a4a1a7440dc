#!/bin/bash
# per-kernel durations of two builds on one box: scripts/ab_fasta_prof.sh ab/lib_old.so ab/lib_new.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp; cd $R
for L in "$@"; do
  O=$R/gpurun_out/abprof_$(basename $L .so); rm -rf $O; mkdir -p $O
  (BLAZESEQ_HIP_LIB=$R/$L timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o ab -- python bench.py --fasta --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0) > $O/kt.log 2>&1 </dev/null
  find $O -type f -size +8M -delete
  echo "== $L"; timeout 60 python scripts/summarize_prof.py $O 2>&1 </dev/null | grep "fa::" | head -8
done
