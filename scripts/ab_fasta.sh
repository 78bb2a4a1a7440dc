#!/bin/bash
# A/B of two builds of the library on one box: scripts/ab_fasta.sh ab/lib_old.so ab/lib_new.so  (FASTA kernels, alternating runs)
for i in 1 2 3; do
  for L in "$@"; do
    BLAZESEQ_HIP_LIB=$PWD/$L timeout 200 python bench.py --fasta --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null </dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', 'ms/step', d['ms_per_step'], 'kernels_ms', d['roofline']['kernels_ms'], 'frac', d['roofline']['frac'])"
  done
done
