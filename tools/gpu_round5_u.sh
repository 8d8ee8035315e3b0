#!/bin/bash
# Round 5, call u: the stream schedule around the column-pair kernel (same box, bench.py headline without legs)
set -u
TAG=${1:-r5u}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
ARGS="--steps 10 --warmup 3 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none"
run() {  # label, env...
  local label=$1; shift
  echo -n "$label : " >> $OUT/ab.txt
  env "$@" timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('value %.0f ms_per_step %.3f factor %.3f frac %.4f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))" >> $OUT/ab.txt
}
for round in 1 2; do
  run "round $round default (two half-batch streams, split diagonal phase)" X=1
  run "round $round one stream                                            " THX_CHOL_SPLIT_MIN=0
  run "round $round three part-batches                                    " THX_CHOL_PARTS=3
  run "round $round fused diagonal kernel                                 " THX_CHOL_SPLIT_DIAG_MIN=1000000
  run "round $round column pairs off                                      " THX_CHOL_COLPAIR=0
done
cat $OUT/ab.txt
