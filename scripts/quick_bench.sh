#!/bin/bash
# quick A/B on the GPU box: parity file + headline bench numbers.  Usage: scripts/quick_bench.sh [bench args]
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'GB/s', d['value'], 'ms', d['roofline_path']['ms'], 'views', d.get('views_mode', {}).get('ms_per_step'), 'fasta', d.get('fasta_mode', {}).get('kernels_ms'))"
