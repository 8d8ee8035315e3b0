#!/bin/bash
# THX_CHOL_LPT (entry-major dispatch of small-batch tile-sparse off-diagonal launches) on / off: BA phases and the chain-graph solver
mkdir -p gpurun_out/r4t
for round in 1 2; do
  for lpt in 0 1; do
    echo "== round $round THX_CHOL_LPT=$lpt" >> gpurun_out/r4t/ab.txt
    THX_CHOL_LPT=$lpt timeout 200 python tools/bench_ba.py 512 8192 256 f32 10 2>&1 | grep -E "^optimize|^phases" | cut -c1-260 >> gpurun_out/r4t/ab.txt
    THX_CHOL_LPT=$lpt timeout 300 python tools/bench_sparse.py 2>&1 | grep -v amdgpu | tail -6 | cut -c1-260 >> gpurun_out/r4t/ab.txt
  done
done
timeout 600 python -m pytest tests/test_gpu_ba.py tests/test_gpu_sparse.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3 >> gpurun_out/r4t/ab.txt
cat gpurun_out/r4t/ab.txt
