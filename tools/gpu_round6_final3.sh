#!/bin/bash
# round 6, last session: the GPU suite + smoke() on the final tree, then MFMA-busy / wave-state counters of the fp64 and fp32 headline factorisations
# (separate --pmc passes, kernel-trace only)
set -u
O=gpurun_out/${1:-r6an}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
G1="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"
G2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
A="--steps 3 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none"
bash tools/pmc.sh r6an_f64 "$G1" "$G2" -- python $(pwd)/bench.py --dtype f64 $A > /dev/null 2>&1
bash tools/pmc.sh r6an_f32 "$G1" "$G2" -- python $(pwd)/bench.py $A > /dev/null 2>&1
for t in r6an_f64 r6an_f32; do echo "#### $t"; grep -A12 "^## chol_offdiag\|^## chol_diag\|^## chol_syrk\|^## chol_potrf" gpurun_out/pmc_$t/summary.txt | grep "^##\|MFMA busy\|wave cycles"; done
