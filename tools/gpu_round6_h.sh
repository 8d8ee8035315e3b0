#!/bin/bash
# bundle adjustment, level mode: schedule knobs of the level factorisation (two half-batch streams, split diagonal phase)
set -u
mkdir -p gpurun_out/r6h
timeout 900 python -m pytest tests/test_gpu_ba.py -m gpu -q -p no:cacheprovider > gpurun_out/r6h/pytest_ba.txt 2>&1; tail -5 gpurun_out/r6h/pytest_ba.txt
run() { echo "== $*" | tee -a gpurun_out/r6h/bench_ba.txt; env "$@" BENCH_BA_ORDERING=nd13 timeout 300 python tools/bench_ba.py 512 8192 256 f32 10 2>&1 | grep "optimize()\|phases" | tee -a gpurun_out/r6h/bench_ba.txt; }
run X=1
run THX_LEVEL_SPLIT_MIN=64
run THX_CHOL_SPLIT_DIAG_MIN=0
run THX_CHOL_SPLIT_DIAG_MIN=0 THX_LEVEL_SPLIT_MIN=64
run THX_CHOL_SPLIT_DIAG_MIN=100000
