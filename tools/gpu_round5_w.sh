#!/bin/bash
# Round 5, call w: MFMA-busy / wave-state counters of the factorisation with column pairs (tools/pmc.sh, kernels serialised) and
# the per-launch executed-flop efficiency on one stream (tools/trace_chol_columns.sh), dense frames and the LM loop's block list.
set -u
TAG=${1:-r5w}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 400 tools/pmc.sh ${TAG}_f32 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" -- python $(pwd)/bench.py --steps 2 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none > $OUT/chol_pmc_f32.txt 2>&1; grep -A10 'chol_offdiag\|chol_syrk\|chol_potrf' $OUT/chol_pmc_f32.txt | head -60
timeout 300 tools/trace_chol_columns.sh ${TAG}_dense 1536 4096 f32 2 > $OUT/cols_dense.txt 2>&1; tail -45 $OUT/cols_dense.txt
THX_COLS_BENCH=1 timeout 300 tools/trace_chol_columns.sh ${TAG}_lm 1536 4096 f32 > $OUT/cols_lm.txt 2>&1; tail -45 $OUT/cols_lm.txt
