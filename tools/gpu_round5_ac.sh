#!/bin/bash
# Round 5, call ac: fp64 off-diagonal kernel (block-compact H) without the 160-byte-per-thread spill of the late panel sub-blocks:
# tests, same-box A/B against the previous library (variants/base.so), WRITE_SIZE / FETCH_SIZE of the fp64 leg.
set -u
TAG=${1:-r5ac}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_block_hessian.py tests/test_gpu_kernels.py tests/test_gpu_lm.py tests/test_gpu_full_size.py -q -x -p no:cacheprovider > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
ARGS="--steps 10 --warmup 3 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none --dtype f64"
for round in 1 2 3; do
  for lib in theseus_amd/lib/variants/base.so ""; do
    echo -n "round $round lib=${lib:-current} f64 : " >> $OUT/ab.txt
    THESEUS_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('value %.0f ms_per_step %.3f factor %.3f frac %.4f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
timeout 500 bash tools/gpu_profile.sh ${TAG}_f64 --dtype f64 --steps 3 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none > $OUT/profile_f64.log 2>&1; grep 'span avg\|factorisation kernels\|chol_offdiag_f64' gpurun_out/prof_${TAG}_f64/summary.txt
