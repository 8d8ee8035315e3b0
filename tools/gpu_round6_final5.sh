#!/bin/bash
# round 6: the right-looking schedule's hand-over by dtype and size (fp32 64 problems / fp64 40 at 12 block columns): GPU suite, smoke, the sweeps with the defaults, the default bench line
set -u
O=gpurun_out/${1:-r6as}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 300 python tools/batch_sweep.py 8,16,32,40,48,64,96,128 2>&1 | grep -v "^$\|amdgpu.ids" | tail -9
timeout 300 python tools/bench_chol.py 3072 48 f32 2>&1 | tail -3
THX_CHOL_RL_MAX_BATCH=32 timeout 300 python tools/bench_chol.py 3072 48 f32 2>&1 | tail -3
timeout 300 python tools/bench_chol.py 2304 40 f32 2>&1 | tail -3
THX_CHOL_RL_MAX_BATCH=32 timeout 300 python tools/bench_chol.py 2304 40 f32 2>&1 | tail -3
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
