#!/bin/bash
# Two PMC passes (SQ instruction mix, then SQ wait / activity) of any command; prints per-kernel averages of the kernels whose
# name matches the filter.  Usage: scripts/gpu_pmc_cmd.sh "<command>" "<grep -E filter>"
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
(timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d $OUT/pmc_sq -o q -- $1) > $OUT/pmc1.log 2>&1
(timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_sq2 -o q -- $1) > $OUT/pmc2.log 2>&1
python scripts/summarize_prof.py $OUT 2>&1 | grep -E -A10 "$2" | grep -v "^--"
