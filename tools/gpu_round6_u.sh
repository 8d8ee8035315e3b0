#!/bin/bash
# subtree streams against ONE launch pair per tree level (THX_LEVEL_SUBTREES=0), same box, interleaved
set -u
mkdir -p gpurun_out/r6u
export BENCH_SPARSE_DENSE=0 BENCH_SPARSE_PHASES=1
for r in 1 2 3; do
for c in 1 0; do
  echo "== round $r THX_LEVEL_SUBTREES=$c  bundle adjustment (auto)" | tee -a gpurun_out/r6u/ab_subtrees.txt
  THX_LEVEL_SUBTREES=$c BENCH_BA_ORDERING=auto timeout 300 python tools/bench_ba.py 512 8192 256 f32 10 2>&1 | grep "optimize()\|phases" | cut -c1-330 | tee -a gpurun_out/r6u/ab_subtrees.txt
done
done
for c in 1 0; do
  echo "== THX_LEVEL_SUBTREES=$c  4096 poses batch 256 ordering nd98" | tee -a gpurun_out/r6u/ab_subtrees.txt
  THX_SPARSE_ORDERING=nd98 THX_LEVEL_SUBTREES=$c timeout 300 python tools/bench_sparse.py 4096 256 f32 40 2>&1 | grep "phases\|sparse:" | cut -c1-300 | tee -a gpurun_out/r6u/ab_subtrees.txt
done
