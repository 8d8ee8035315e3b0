#!/bin/bash
# Round 4, first measurement set: why lm_trunc differs (tools/diag_lm_trunc.py), the full GPU suite, the drop-in on hardware.
set -u
TAG=${1:-r4b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 200 python tools/diag_lm_trunc.py lm_trunc > $OUT/diag_lm_trunc.txt 2>&1; head -3 $OUT/diag_lm_trunc.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
if [ -d _refcopy ]; then
  export THX_REFERENCE_ROOT=$(pwd)/_refcopy THX_PLUGIN_DEVICE=cuda
  timeout 600 python -m pytest tests/test_plugin_reference.py -q --maxfail=20 -p no:cacheprovider > $OUT/pytest_plugin_cuda.txt 2>&1; tail -5 $OUT/pytest_plugin_cuda.txt
  timeout 600 python tools/dropin_bench.py --steps 10 > $OUT/dropin_bench_f32.log 2>&1; tail -1 $OUT/dropin_bench_f32.log
  timeout 600 python tools/dropin_bench.py --steps 10 --adaptive > $OUT/dropin_bench_f32_adaptive.log 2>&1; tail -1 $OUT/dropin_bench_f32_adaptive.log
fi
