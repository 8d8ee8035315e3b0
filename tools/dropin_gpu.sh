#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the real theseus x libtheseus_hip.so.  Expects a scratch copy of the reference's Python
# packages under ./_refcopy (made by tools/dropin_stage.sh in the build container, git-ignored, removed after the call).
# Output: gpurun_out/dropin/*.log
set -u
OUT=gpurun_out/dropin; mkdir -p $OUT
export THX_REFERENCE_ROOT=$(pwd)/_refcopy THX_PLUGIN_DEVICE=cuda
python -m pytest tests/test_plugin_reference.py -q -x -p no:cacheprovider > $OUT/pytest_plugin_cuda.log 2>&1
tail -3 $OUT/pytest_plugin_cuda.log
timeout 900 python tools/dropin_bench.py --steps 10 > $OUT/dropin_bench_f32.log 2>&1;            tail -1 $OUT/dropin_bench_f32.log
python tools/dropin_bench.py --steps 10 --no-hooks > $OUT/dropin_bench_f32_nohooks.log 2>&1; tail -1 $OUT/dropin_bench_f32_nohooks.log
python tools/dropin_bench.py --steps 10 --adaptive > $OUT/dropin_bench_f32_adaptive.log 2>&1; tail -1 $OUT/dropin_bench_f32_adaptive.log
python tools/dropin_bench.py --steps 5 --dtype f64 --batch 2048 > $OUT/dropin_bench_f64.log 2>&1; tail -1 $OUT/dropin_bench_f64.log
