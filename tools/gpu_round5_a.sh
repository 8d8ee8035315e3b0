#!/bin/bash
# Round 5, call a: upper bound of running the batch as time-offset part-batches on their own streams (tools/exp_pipeline_parts.py).
set -u
TAG=${1:-r5a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
R=$OUT/exp_pipeline_parts.txt
run() { echo "# $*" >> $R; ( "$@" ) >> $R 2>> $OUT/err.txt; }
run timeout 200 python tools/exp_pipeline_parts.py --parts 1
THX_CHOL_SPLIT_MIN=0 run timeout 200 python tools/exp_pipeline_parts.py --parts 1
echo "# THX_CHOL_SPLIT_MIN=0 below" >> $R
THX_CHOL_SPLIT_MIN=0 run timeout 200 python tools/exp_pipeline_parts.py --parts 2 --offset 0
THX_CHOL_SPLIT_MIN=0 run timeout 200 python tools/exp_pipeline_parts.py --parts 2 --offset 1
THX_CHOL_SPLIT_MIN=0 run timeout 200 python tools/exp_pipeline_parts.py --parts 4 --offset 1
echo "# default split below" >> $R
run timeout 200 python tools/exp_pipeline_parts.py --parts 2 --offset 1
run timeout 200 python tools/exp_pipeline_parts.py --parts 2 --offset 0
echo "# fp64" >> $R
run timeout 300 python tools/exp_pipeline_parts.py --parts 1 --dtype f64 --iters 10
THX_CHOL_SPLIT_MIN=0 run timeout 300 python tools/exp_pipeline_parts.py --parts 2 --offset 1 --dtype f64 --iters 10
run timeout 300 python tools/exp_pipeline_parts.py --parts 2 --offset 1 --dtype f64 --iters 10
cat $R; tail -5 $OUT/err.txt
