#!/bin/bash
# A/B: the factorisation's batch split over 2 (default) vs 3 staggered streams.  usage: tools/ab_parts.sh <tag>
set -u
TAG=${1:-abparts}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none"
for round in 1 2; do
  for parts in 2 3; do
    for dt in f32 f64; do
      echo -n "round $round parts=$parts $dt : " >> $OUT/ab_parts.txt
      THX_CHOL_PARTS=$parts timeout 300 python bench.py $ARGS --dtype $dt 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('value %.0f ms_per_step %.3f factor %.3f frac %.4f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))" >> $OUT/ab_parts.txt
    done
  done
done
cat $OUT/ab_parts.txt
