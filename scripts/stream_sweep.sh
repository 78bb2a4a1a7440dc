#!/bin/bash
# k_stream (EXPERIMENTS build) over super-tile size and look-back group size, on one box.  Usage: scripts/stream_sweep.sh [out]
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=${1:-$R/gpurun_out/stream_sweep.txt}; cd $R/blazeseq_amd/csrc
for st in 2 3 4; do for g in 16 32 64; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-bitwise-instead-of-logical -DBZQ_EXPERIMENTS=1 -DBZQ_STREAM_ST=$st -DBZQ_STREAM_SGRP=$g -c -o /tmp/api_$st_$g.o bzq_api.hip 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/libexp_${st}_${g}.so /tmp/api_$st_$g.o build_exp/bzq_fasta.o -lz -ldl -lpthread -lrt &&
  (cd $R && BLAZESEQ_HIP_LIB=/tmp/libexp_${st}_${g}.so python bench.py --no-cpu-baseline --steps 10 --warmup 3 --min-seconds 0.2 --stream --ablate 64 2>/tmp/err_${st}_${g}.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ST=$st SGRP=$g ms/step', d['ms_per_step'], 'emit', d['roofline_path']['ms']['emit'])"; grep phase_cycles /tmp/err_${st}_${g}.txt | tail -1)
done; done 2>&1 | tee $OUT
