#!/bin/bash
set -u
mkdir -p gpurun_out/r6b
export BENCH_SPARSE_DENSE=0 BENCH_SPARSE_PHASES=1
for cfg in "4096 64" "4096 256" "1024 64" "4096 8" "1024 256"; do
  set -- $cfg
  python tools/bench_sparse.py $1 $2 f32 5 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6b/bench_sparse.txt
done
THX_SPARSE_ORDERING=rcm python tools/bench_sparse.py 4096 64 f32 5 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6b/bench_sparse.txt
