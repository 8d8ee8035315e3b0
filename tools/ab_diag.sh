#!/bin/bash
# A/B of the diagonal phase of the Cholesky: split (chol_syrk + chol_potrf, default) vs fused (chol_diag, THX_CHOL_FUSED_DIAG=1).
# usage: tools/ab_diag.sh <tag>     (outputs under gpurun_out/<tag>/)
set -u
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for cfg in "1536 4096 f32 3" "1536 4096 f64 2" "1536 1024 f32 3" "1536 256 f32 5" "3072 256 f32 3" "1536 1024 f64 3"; do
  for fused in 0 1; do
    echo "== n B dtype reps = $cfg ; THX_CHOL_FUSED_DIAG=$fused" >> $OUT/ab_diag.txt
    THX_CHOL_FUSED_DIAG=$fused timeout 300 python tools/bench_chol.py $cfg 2>&1 | grep -v amdgpu >> $OUT/ab_diag.txt
  done
done
cat $OUT/ab_diag.txt
