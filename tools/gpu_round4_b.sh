#!/bin/bash
# Round 4: look-ahead schedule of the small-batch factorisation (THX_CHOL_LOOKAHEAD), same box A/B + the tests that pin bit-identity.
set -u
TAG=${1:-r4c}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 400 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_ba.py tests/test_gpu_kernels.py tests/test_gpu_block_hessian.py tests/test_gpu_unrolled.py -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/pytest_gpu_subset.txt 2>&1; tail -4 $OUT/pytest_gpu_subset.txt
for la in 0 1 0 1; do
  echo "== THX_CHOL_LOOKAHEAD=$la ==" >> $OUT/ab_lookahead.txt
  THX_CHOL_LOOKAHEAD=$la timeout 200 python tools/bench_ba.py 2>&1 | grep -v amdgpu | tail -22 >> $OUT/ab_lookahead.txt
  for cfg in "1536 256 f32" "1536 512 f32" "3072 256 f32" "1536 256 f64"; do
    echo "-- bench_chol $cfg" >> $OUT/ab_lookahead.txt
    THX_CHOL_LOOKAHEAD=$la timeout 100 python tools/bench_chol.py $cfg 5 2>&1 | grep -v amdgpu | tail -4 >> $OUT/ab_lookahead.txt
  done
done
cat $OUT/ab_lookahead.txt | grep -E "==|--|chol_factor|factor|per " | head -80
