#!/bin/bash
# Round 5, call af (final tree): kernel trace + HBM counters of the headline bench (tools/gpu_profile.sh) and the MFMA-busy / wave-state
# counters of the fp64 factorisation after the spill fix (tools/pmc.sh).
set -u
TAG=${1:-r5af}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 600 bash tools/gpu_profile.sh ${TAG} > $OUT/profile.log 2>&1; head -8 gpurun_out/prof_${TAG}/summary.txt; grep 'span avg\|factorisation kernels' gpurun_out/prof_${TAG}/summary.txt
timeout 500 tools/pmc.sh ${TAG}_f64 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" -- python $(pwd)/bench.py --dtype f64 --steps 2 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none > $OUT/chol_pmc_f64.txt 2>&1; grep -A10 'chol_offdiag\|chol_syrk\|chol_potrf' $OUT/chol_pmc_f64.txt | grep 'chol_\|MFMA busy\|wave cycles'
