#!/bin/bash
set -u
mkdir -p gpurun_out/r6c
python -m pytest tests/test_gpu_sparse.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -5
export BENCH_SPARSE_DENSE=0 BENCH_SPARSE_PHASES=1
for cfg in "4096 64" "4096 256" "1024 64" "4096 8"; do
  set -- $cfg
  python tools/bench_sparse.py $1 $2 f32 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6c/bench_sparse.txt
done
THX_LEVEL_SPLIT_MIN=0 python tools/bench_sparse.py 4096 256 f32 10 2>&1 | grep -v amdgpu.ids | sed 's/^/[one stream] /' | tee -a gpurun_out/r6c/bench_sparse.txt
THX_LEVEL_SPLIT_MIN=0 python tools/bench_sparse.py 4096 64 f32 10 2>&1 | grep -v amdgpu.ids | sed 's/^/[one stream] /' | tee -a gpurun_out/r6c/bench_sparse.txt
