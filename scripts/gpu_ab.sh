#!/bin/bash
# A/B of library builds on the SAME box (box-to-box variation is ~5%): scripts/gpu_ab.sh libA.so libB.so [bench args]
A=$1; B=$2; shift 2
for i in 1 2 3; do for l in $A $B; do
  BLAZESEQ_HIP_LIB=$PWD/$l python bench.py --no-cpu-baseline $* 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'], d['roofline_path']['ms'])"
done; done
