#!/bin/bash
# Round 6 baseline: tile-sparse solver (RCM chain) on the reference's sweep sizes, kernel stats
set -u
mkdir -p gpurun_out/r6a
export BENCH_SPARSE_DENSE=0
for cfg in "4096 64" "4096 256" "1024 64" "1024 256"; do
  set -- $cfg
  python tools/bench_sparse.py $1 $2 f32 5 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6a/bench_sparse.txt
done
tools/kernel_stats.sh gpurun_out/r6a/kstats_4096_b64.txt -- python tools/bench_sparse.py 4096 64 f32 5
tools/kernel_stats.sh gpurun_out/r6a/kstats_4096_b256.txt -- python tools/bench_sparse.py 4096 256 f32 5
