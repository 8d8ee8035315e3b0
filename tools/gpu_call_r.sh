#!/bin/bash
set -u
OUT=gpurun_out/dropin; mkdir -p $OUT
export THX_REFERENCE_ROOT=$(pwd)/_refcopy THX_PLUGIN_DEVICE=cuda
timeout 900 python tools/dropin_bench.py --steps 10 > $OUT/dropin_bench_f32.log 2>&1; tail -1 $OUT/dropin_bench_f32.log
timeout 900 python tools/dropin_bench.py --steps 5 --dtype f64 --batch 2048 > $OUT/dropin_bench_f64.log 2>&1; tail -1 $OUT/dropin_bench_f64.log
