#!/bin/bash
# Round 5, call ab: kernel trace + HBM counters of the fp64 and bundle-adjustment legs with the round's kernels (tools/gpu_profile.sh)
set -u
TAG=${1:-r5ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 500 bash tools/gpu_profile.sh ${TAG}_f64 --dtype f64 --steps 3 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none > $OUT/profile_f64.log 2>&1; head -12 gpurun_out/prof_${TAG}_f64/summary.txt; grep 'span avg\|factorisation kernels' gpurun_out/prof_${TAG}_f64/summary.txt
timeout 700 bash tools/gpu_profile.sh ${TAG}_ba --steps 2 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs ba > $OUT/profile_ba.log 2>&1; head -14 gpurun_out/prof_${TAG}_ba/summary.txt
