#!/bin/bash
# Round 4: counters for configs 3 (fp64) and 4 (bundle adjustment) -- kernel trace + FETCH_SIZE / WRITE_SIZE passes of bench.py --,
# the full GPU suite on the current tree, the drop-in with and without the lagged failure check.
set -u
TAG=${1:-r4d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 500 bash tools/gpu_profile.sh ${TAG}_f64 --dtype f64 --steps 3 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none > $OUT/profile_f64.log 2>&1; tail -30 gpurun_out/prof_${TAG}_f64/summary.txt
timeout 700 bash tools/gpu_profile.sh ${TAG}_ba --steps 2 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs ba > $OUT/profile_ba.log 2>&1; tail -40 gpurun_out/prof_${TAG}_ba/summary.txt
if [ -d _refcopy ]; then
  export THX_REFERENCE_ROOT=$(pwd)/_refcopy THX_PLUGIN_DEVICE=cuda
  for extra in "" "--lagged" "--adaptive" "--adaptive --lagged"; do
    timeout 300 python tools/dropin_bench.py --steps 10 $extra > $OUT/dropin.log 2>&1; tail -1 $OUT/dropin.log >> $OUT/dropin_bench.txt
  done
  cat $OUT/dropin_bench.txt | cut -c1-420
fi
