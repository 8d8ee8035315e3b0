#!/bin/bash
# factor / solve time of the level schedule under different dissection depths (calibration of the ordering's time model)
set -u
mkdir -p gpurun_out/r6d
export BENCH_SPARSE_DENSE=0 BENCH_SPARSE_PHASES=1
for B in 8 64 256; do
for m in nd1 nd13 nd25 nd49 nd98 band rcm; do
  echo "== batch $B ordering $m" | tee -a gpurun_out/r6d/orderings.txt
  THX_SPARSE_ORDERING=$m python tools/bench_sparse.py 4096 $B f32 10 2>&1 | grep -v amdgpu.ids | grep "phases\|sparse:" | tee -a gpurun_out/r6d/orderings.txt
done
done
