#!/bin/bash
# block-row backward solve (chol_bwd_rows_kernel) against the one-workgroup-per-problem kernel: tests, then the LM iteration at
# batch 8 ... 256 with it off / on (separate processes: the threshold is read once at load time)
set -u
mkdir -p gpurun_out/r6o
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sparse.py tests/test_gpu_lm.py tests/test_gpu_full_size.py tests/test_gpu_implicit.py -m gpu -q -p no:cacheprovider > gpurun_out/r6o/pytest.txt 2>&1; tail -6 gpurun_out/r6o/pytest.txt
for v in 0 100000; do
  echo "== THX_CHOL_BWD_ROWS_MAX_BATCH=$v" | tee -a gpurun_out/r6o/ab_bwd_rows.txt
  THX_CHOL_BWD_ROWS_MAX_BATCH=$v timeout 600 python tools/ab_small_batch.py 8,16,32,64,128,256 2>&1 | grep -v "amdgpu.ids\|no pairs" | tee -a gpurun_out/r6o/ab_bwd_rows.txt
done
