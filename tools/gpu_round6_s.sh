#!/bin/bash
set -u
mkdir -p gpurun_out/r6s
export BENCH_SPARSE_DENSE=0 BENCH_SPARSE_PHASES=1
for o in nd98 nd1; do
for c in 1 0; do
  echo "== THX_LEVEL_CHAINS=$c  4096 poses batch 256 ordering $o" | tee -a gpurun_out/r6s/ab_chains.txt
  THX_SPARSE_ORDERING=$o THX_LEVEL_CHAINS=$c timeout 300 python tools/bench_sparse.py 4096 256 f32 40 2>&1 | grep "phases\|sparse:\|poses /" | cut -c1-300 | tee -a gpurun_out/r6s/ab_chains.txt
done
done
for B in 8 64; do
  echo "== default 4096 poses batch $B" | tee -a gpurun_out/r6s/ab_chains.txt
  timeout 300 python tools/bench_sparse.py 4096 $B f32 40 2>&1 | grep "phases\|sparse:\|poses /" | cut -c1-300 | tee -a gpurun_out/r6s/ab_chains.txt
done
