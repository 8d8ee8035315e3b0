#!/bin/bash
# Same-box A/B of the headline iteration: block-compact vs dense Hessian, split vs fused diagonal phase (two rounds, interleaved).
# usage: tools/ab_bench.sh <tag> [bench args]
set -u
TAG=${1:-ab}; shift || true
OUT=gpurun_out/$TAG; mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none $@"
for round in 1 2; do
  for cfg in "0 2048" "1 2048" "0 1000000" "1 1000000"; do
    set -- $cfg
    echo -n "round $round dense_hessian=$1 split_diag_min=$2 : " >> $OUT/ab_bench.txt
    THX_DENSE_HESSIAN=$1 THX_CHOL_SPLIT_DIAG_MIN=$2 timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
p = r['phases_ms_per_call']
print('value %.0f ms_per_step %.3f factor %.3f assemble %.3f bwd %.3f frac %.4f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], p.get('pg_assemble', 0), p.get('chol_solve_backward', 0), r['roofline']['frac']))" >> $OUT/ab_bench.txt
  done
done
cat $OUT/ab_bench.txt
