#!/bin/bash
# Round 5, call s: column pairs (chol_offdiag2_f32_kernel): bit-identity tests, then same-box A/B on the dense-frame factor
# (bench_chol) and inside the LM loop (bench.py headline, block-compact H).
set -u
TAG=${1:-r5s}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_block_hessian.py -q -x -p no:cacheprovider -k "column_pairs or factor_from_blocks or chol_factor_solve_vs_lapack or agree" > $OUT/pytest_pairs.txt 2>&1; tail -4 $OUT/pytest_pairs.txt
ARGS="--steps 10 --warmup 3 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none"
for round in 1 2; do
  for cp in 0 1; do
    echo -n "round $round THX_CHOL_COLPAIR=$cp bench_chol dense H : " >> $OUT/ab.txt
    THX_CHOL_COLPAIR=$cp timeout 300 python tools/bench_chol.py 1536 4096 f32 5 2>&1 | grep -E "^n=" | sed 's/; solve.*//' >> $OUT/ab.txt
    echo -n "round $round THX_CHOL_COLPAIR=$cp bench.py LM loop     : " >> $OUT/ab.txt
    THX_CHOL_COLPAIR=$cp timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('value %.0f ms_per_step %.3f factor %.3f frac %.4f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
