#!/bin/bash
# same-box A/B of a ctx option: scripts/ab_opt.sh fold_rebase 0 1 [bench args]   (box-to-box variation is ~5 %: only compare inside one call)
K=$1; A=$2; B=$3; shift 3
for i in 1 2 3; do for v in $A $B; do
  python bench.py --no-cpu-baseline --no-extra-modes --opt $K=$v $* 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$K=$v', d['value'], d['ms_per_step'], d['roofline_path']['ms'])"
done; done
