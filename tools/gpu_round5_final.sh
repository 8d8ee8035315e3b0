#!/bin/bash
# Round 5's measurement set in ONE gpurun call (outputs under gpurun_out/<tag>/; copy what is to be judged into profiles/r5/): the
# GPU test suite, the default bench.py line (headline + all legs) the way the driver runs it, and -- when a scratch copy of the
# reference was staged (tools/dropin_stage.sh) -- the plugin's tests on the HIP kernels and the drop-in leg of bench.py.
# Every command under its own timeout.
set -u
TAG=${1:-r5z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
tail -3 $OUT/bench_default.time; wc -c $OUT/bench_default.json
if [ -d _refcopy ]; then
  export THX_REFERENCE_ROOT=$(pwd)/_refcopy THX_PLUGIN_DEVICE=cuda
  timeout 600 python -m pytest tests/test_plugin_reference.py -q --maxfail=20 -p no:cacheprovider > $OUT/pytest_plugin_cuda.txt 2>&1; tail -3 $OUT/pytest_plugin_cuda.txt
  timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs dropin > $OUT/bench_dropin_leg.json 2> $OUT/bench_dropin_leg.err
  wc -c $OUT/bench_dropin_leg.json
fi
