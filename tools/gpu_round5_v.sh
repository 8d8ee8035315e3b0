#!/bin/bash
# Round 5, call v: branch-free prefetch of the block-compact Hessian's pieces (HBPre::load, HBPre2 in the pair kernel): bit-identity
# tests of every path that reads the block list, same-box A/B against the previous library (theseus_amd/lib/variants/base.so).
set -u
TAG=${1:-r5v}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_block_hessian.py tests/test_gpu_kernels.py tests/test_gpu_sparse.py tests/test_gpu_lm.py tests/test_gpu_full_size.py -q -x -p no:cacheprovider > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
ARGS="--steps 10 --warmup 3 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none"
for round in 1 2; do
  for lib in theseus_amd/lib/variants/base.so ""; do
    for dt in f32 f64; do
      echo -n "round $round lib=${lib:-current} $dt : " >> $OUT/ab.txt
      THESEUS_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py $ARGS --dtype $dt 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('value %.0f ms_per_step %.3f factor %.3f frac %.4f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))" >> $OUT/ab.txt
    done
  done
done
cat $OUT/ab.txt
