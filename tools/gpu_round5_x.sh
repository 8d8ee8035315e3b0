#!/bin/bash
# Round 5, call x: (1) bench.py's multi-rank path with the round's kernels: two ranks on one GPU over gloo (tools/preflight_2ranks_one_gpu.sh);
# (2) fp64: per-launch executed-flop efficiency on one stream + socket power / clock samples while bench.py --dtype f64 runs.
set -u
TAG=${1:-r5x}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 700 bash tools/preflight_2ranks_one_gpu.sh > $OUT/preflight.txt 2>&1; tail -4 $OUT/preflight.txt; cp gpurun_out/preflight/line.json $OUT/preflight_line.json 2>/dev/null
THX_COLS_BENCH=1 timeout 300 tools/trace_chol_columns.sh ${TAG}_f64 1536 4096 f64 > $OUT/cols_f64.txt 2>&1; grep -v '^[EW]2026' $OUT/cols_f64.txt | tail -42
