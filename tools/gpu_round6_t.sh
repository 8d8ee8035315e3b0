#!/bin/bash
set -u
mkdir -p gpurun_out/r6t
export BENCH_SPARSE_DENSE=0 BENCH_SPARSE_PHASES=1
for c in 1 0; do
  echo "== THX_LEVEL_CHAINS=$c  bundle adjustment (auto)" | tee -a gpurun_out/r6t/ab_chains.txt
  THX_LEVEL_CHAINS=$c BENCH_BA_ORDERING=auto timeout 300 python tools/bench_ba.py 512 8192 256 f32 10 2>&1 | grep "optimize()\|phases" | cut -c1-330 | tee -a gpurun_out/r6t/ab_chains.txt
done
for c in 1 0; do
  echo "== THX_LEVEL_CHAINS=$c  4096 poses batch 256 ordering nd98" | tee -a gpurun_out/r6t/ab_chains.txt
  THX_SPARSE_ORDERING=nd98 THX_LEVEL_CHAINS=$c timeout 300 python tools/bench_sparse.py 4096 256 f32 40 2>&1 | grep "phases\|sparse:" | cut -c1-300 | tee -a gpurun_out/r6t/ab_chains.txt
done
timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_ba.py tests/test_gpu_full_size.py tests/test_gpu_unrolled.py -m gpu -q -p no:cacheprovider > gpurun_out/r6t/pytest.txt 2>&1; tail -4 gpurun_out/r6t/pytest.txt
