#!/bin/bash
# What bounds the emit kernel (k_fused<LB=false>)?  One table, on one box:
#   the kernel as shipped / with its stores confined to 1 MiB (L2-resident: no HBM writes) / without the scatter / stopped
#   after the tile is staged; the same on a 95 MB input that the previous launch left in the Infinity Cache; a pure copy
#   kernel of the same shape (scripts/probes/hop_probe method 0); and the LDS counters of the full kernel.
# Needs the EXPERIMENTS build (timing switches are compiled out of the product).  Usage: scripts/emit_bound.sh [outdir]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/emit_bound}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
export BLAZESEQ_HIP_LIB=$R/blazeseq_amd/libblazeseq_hip_exp.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['roofline_path']['ms']; print('$1', 'reads', d['config']['records_per_gpu'], 'ms/step', d['ms_per_step'], 'aggregate', m['aggregate'], 'emit', m['emit'], 'rebase', m['rebase'])"; }
for ab in 0 8 16 2 4 1 128 256 512 1024; do
  python bench.py --no-cpu-baseline --steps 20 --warmup 3 --ablate $ab 2>/dev/null | line "ablate=$ab 10M"
done | tee $OUT/ablate_10M.txt
for ab in 0 8 1; do
  python bench.py --no-cpu-baseline --steps 200 --warmup 20 --reads 300000 --ablate $ab 2>/dev/null | line "ablate=$ab 300k(95MB,MALL-resident)"
done | tee $OUT/ablate_300k.txt
for pb in 67108864 134217728; do
  python bench.py --no-cpu-baseline --steps 20 --warmup 3 --pass-bytes $pb 2>/dev/null | line "pass_bytes=$pb 10M"
done | tee $OUT/pass_bytes.txt
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/hop_probe scripts/probes/hop_probe.hip && /tmp/hop_probe | tee $OUT/hop_probe.txt
for ab in 0 8; do
  (timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d $OUT/pmc_lds_$ab -o lds -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 --min-seconds 0 --ablate $ab) > $OUT/pmc_lds_$ab.log 2>&1
  python scripts/summarize_pmc.py $OUT/pmc_lds_$ab k_fused | tee $OUT/pmc_lds_$ab.txt
done
find $OUT -type f -size +8M -delete
