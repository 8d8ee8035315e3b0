#!/bin/bash
# round 6: look-ahead inside the right-looking schedule of small dense batches (THX_CHOL_RL_LOOKAHEAD, default on): diag(j) takes the previous column's update of its
# own tile (a one-tile K-loop), the trailing update of column j - 1 runs beside it on a second stream
O=gpurun_out/${1:-r6rlla}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_full_size.py tests/test_gpu_lm.py tests/test_gpu_block_hessian.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15
for rep in 1 2; do
for v in 1 0; do
  export THX_CHOL_RL_LOOKAHEAD=$v
  echo "== RL look-ahead $v round $rep"
  timeout 300 python tools/batch_sweep.py 8,16,32 2>&1 | grep -v "^$\|amdgpu.ids" | tail -4
  timeout 300 python tools/ab_small_batch.py 8,16,32 f64 2>&1 | grep "right-looking"
  if [ $rep = 1 ]; then
    for b in 8 32; do timeout 300 python tools/bench_chol.py 1536 $b f32 2>&1 | tail -3; done
    timeout 300 python tools/bench_chol.py 3072 16 f32 2>&1 | tail -3
    timeout 300 python tools/bench_chol.py 1290 8 f64 2>&1 | tail -3
  fi
done; done
