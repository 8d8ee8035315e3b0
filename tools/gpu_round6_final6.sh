#!/bin/bash
# round 6: mode 1 of the right-looking schedule with block column 1 folded in too (no separate update(0) launch): GPU suite, smoke, sweeps, default bench line
set -u
O=gpurun_out/${1:-r6au}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 300 python tools/batch_sweep.py 8,16,32,40,48,64 2>&1 | grep -v "^$\|amdgpu.ids" | tail -7
timeout 300 python tools/ab_small_batch.py 8,32 f64 2>&1 | grep "right-looking"
THX_CHOL_RL_LOOKAHEAD=1 timeout 300 python tools/ab_small_batch.py 8,32 f64 2>&1 | grep "right-looking"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
