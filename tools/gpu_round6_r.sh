#!/bin/bash
# subtree streams of the level schedule (thx_level_schedule.level_stream_host): on / off (THX_LEVEL_CHAINS), bundle adjustment
# and the 4096-pose chain graph; then the sparse / BA GPU tests
set -u
mkdir -p gpurun_out/r6r
export BENCH_SPARSE_DENSE=0 BENCH_SPARSE_PHASES=1
for c in 1 0 1 0; do
  echo "== THX_LEVEL_CHAINS=$c  bundle adjustment (nd13)" | tee -a gpurun_out/r6r/ab_chains.txt
  THX_LEVEL_CHAINS=$c BENCH_BA_ORDERING=auto timeout 300 python tools/bench_ba.py 512 8192 256 f32 10 2>&1 | grep "optimize()\|phases\|level mode" | cut -c1-330 | tee -a gpurun_out/r6r/ab_chains.txt
done
for B in 8 64 256; do
for c in 1 0; do
  echo "== THX_LEVEL_CHAINS=$c  4096 poses batch $B" | tee -a gpurun_out/r6r/ab_chains.txt
  THX_LEVEL_CHAINS=$c timeout 300 python tools/bench_sparse.py 4096 $B f32 40 2>&1 | grep "phases\|sparse:\|poses /" | cut -c1-300 | tee -a gpurun_out/r6r/ab_chains.txt
done
done
timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_ba.py tests/test_gpu_full_size.py -m gpu -q -p no:cacheprovider > gpurun_out/r6r/pytest.txt 2>&1; tail -6 gpurun_out/r6r/pytest.txt
