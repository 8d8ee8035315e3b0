#!/bin/bash
# Round 6: kernel trace + HBM counters (separate --pmc passes) of the headline and of the bundle-adjustment leg on the round's tree
set -u
TAG=${1:-r6n}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 600 bash tools/gpu_profile.sh ${TAG}_f32 --steps 3 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none > $OUT/profile_f32.log 2>&1; head -14 gpurun_out/prof_${TAG}_f32/summary.txt; grep 'span avg\|factorisation kernels' gpurun_out/prof_${TAG}_f32/summary.txt
timeout 700 bash tools/gpu_profile.sh ${TAG}_ba --steps 2 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs ba > $OUT/profile_ba.log 2>&1; head -16 gpurun_out/prof_${TAG}_ba/summary.txt
