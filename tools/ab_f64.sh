#!/bin/bash
# fp64 A/B: k-chunks of 32 columns (current build) vs 16 (theseus_amd/lib/variants/kb16.so), bench_chol + the bench's fp64 iteration.
set -u
TAG=${1:-abf64}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do
  for lib in "" "theseus_amd/lib/variants/kb16.so"; do
    for cfg in "1536 4096 f64 2" "1536 1024 f64 3" "3072 256 f64 2"; do
      echo "== round $round lib=${lib:-current} n B dtype reps = $cfg" >> $OUT/ab_f64.txt
      THESEUS_HIP_LIB=${lib:+$(pwd)/$lib} timeout 300 python tools/bench_chol.py $cfg 2>&1 | grep -v amdgpu | grep -E "^n=|resid" >> $OUT/ab_f64.txt
    done
    echo -n "== round $round lib=${lib:-current} bench.py --dtype f64: " >> $OUT/ab_f64.txt
    THESEUS_HIP_LIB=${lib:+$(pwd)/$lib} timeout 300 python bench.py --dtype f64 --steps 10 --warmup 2 --cpu-sample 0 --parity-sample 8 --no-sparse-leg --legs none 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('value %.0f ms_per_step %.3f factor %.3f frac %.4f parity %.2e' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['parity']['hip_max_abs_pose_err']))" >> $OUT/ab_f64.txt
  done
done
cat $OUT/ab_f64.txt
