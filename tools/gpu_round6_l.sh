#!/bin/bash
# per-call vs per-iteration cost of the level-scheduled sparse solver: the same problem at 10 and 40 LM iterations; host profile
set -u
mkdir -p gpurun_out/r6l
export BENCH_SPARSE_DENSE=0
for B in 8 64; do
for it in 10 40; do
  echo "== batch $B iters $it" | tee -a gpurun_out/r6l/slope.txt
  python tools/bench_sparse.py 4096 $B f32 $it 2>&1 | grep "timed\|sparse:" | tee -a gpurun_out/r6l/slope.txt
done
done
python -m cProfile -s tottime tools/bench_sparse.py 4096 8 f32 40 2>&1 | grep -v amdgpu | head -60 > gpurun_out/r6l/cprofile_b8_40it.txt
