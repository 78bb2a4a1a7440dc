#!/bin/bash
# A/B/C of library builds on the SAME box, interleaved: scripts/gpu_ab3.sh "libA.so libB.so libC.so" [bench args]
LIBS=$1; shift
for i in 1 2 3; do for l in $LIBS; do
  BLAZESEQ_HIP_LIB=$PWD/$l python bench.py --no-cpu-baseline --no-extra-modes $* 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'], d['roofline_path']['ms'])"
done; done
