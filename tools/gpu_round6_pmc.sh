#!/bin/bash
# MFMA-busy + wave-state counters (separate --pmc passes, kernel-trace only) of the round's new schedules: bundle adjustment in
# level mode, the 4096-pose chain graph at batch 64, the right-looking dense schedule at batch 8 / 32
set -u
export BENCH_SPARSE_DENSE=0
G1="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"
G2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
bash tools/pmc.sh r6_ba_mfma "$G1" "$G2" -- python $(pwd)/tools/bench_ba.py 512 8192 256 f32 3 > /dev/null 2>&1
bash tools/pmc.sh r6_sparse_b64_mfma "$G1" "$G2" -- python $(pwd)/tools/bench_sparse.py 4096 64 f32 5 > /dev/null 2>&1
bash tools/pmc.sh r6_rl_mfma "$G1" "$G2" -- python $(pwd)/tools/ab_small_batch.py 8,32 > /dev/null 2>&1
for t in r6_ba_mfma r6_sparse_b64_mfma r6_rl_mfma; do echo "#### $t"; grep -A12 "^## chol_offdiag\|^## chol_diag\|^## chol_syrk\|^## chol_potrf" gpurun_out/pmc_$t/summary.txt | grep "^##\|MFMA busy\|wave cycles"; done
