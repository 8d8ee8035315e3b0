#!/bin/bash
# round 6, end of the last session: GPU suite + smoke() + the default bench line on the FINAL tree (right-looking schedule: two launches per block column in fp32,
# the second stream in fp64), then the small-batch sweeps of both dtypes with the default modes
set -u
O=gpurun_out/${1:-r6aq}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
timeout 300 python tools/batch_sweep.py 8,16,32,64,256 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6
timeout 300 python tools/ab_small_batch.py 8,16,32 f64 2>&1 | grep "right-looking"
