#!/bin/bash
# Round 5, call ag (final tree, no code change): fp64 after the spill fix -- per-launch executed-flop efficiency on one stream and the
# socket power / clock of the fp64 and fp32 factorisations (tools/power_model.py).
set -u
TAG=${1:-r5ag}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
THX_COLS_BENCH=1 timeout 300 tools/trace_chol_columns.sh ${TAG}_f64 1536 4096 f64 > $OUT/cols_f64.txt 2>&1; grep -v '^[EW]2026' $OUT/cols_f64.txt | tail -38
timeout 300 python tools/power_model.py 4 idle,f32,f64 2>&1 | grep -v amdgpu.ids > $OUT/power_model.txt; cat $OUT/power_model.txt
