#!/bin/bash
# Round 5, call aa: batch 1024 (config 5's size) with column pairs: stream schedule / diagonal-phase thresholds (same box)
set -u
TAG=${1:-r5aa}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
run() {  # label, batch, env...
  local label=$1 batch=$2; shift 2
  echo -n "$label : " >> $OUT/ab.txt
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --batch $batch --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('value %.0f ms_per_step %.3f factor %.3f frac %.4f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))" >> $OUT/ab.txt
}
for round in 1 2; do
  for B in 1024 2048; do
    run "round $round batch $B default                         " $B X=1
    run "round $round batch $B one stream                      " $B THX_CHOL_SPLIT_MIN=0
    run "round $round batch $B split diagonal phase            " $B THX_CHOL_SPLIT_DIAG_MIN=0
    run "round $round batch $B one stream, split diagonal phase" $B THX_CHOL_SPLIT_MIN=0 THX_CHOL_SPLIT_DIAG_MIN=0
    run "round $round batch $B column pairs off                " $B THX_CHOL_COLPAIR=0
  done
done
cat $OUT/ab.txt
