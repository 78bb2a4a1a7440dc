#!/bin/bash
# scripts/gpu_ab_multi.sh "<bench args>" lib1.so lib2.so ...   (2 rounds, same box)
ARGS=$1; shift
for i in 1 2; do for l in "$@"; do
  BLAZESEQ_HIP_LIB=$PWD/$l python bench.py --no-cpu-baseline $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'], d['roofline_path']['ms'])"
done; done
