#!/bin/bash
# Round 6's measurement set in ONE gpurun call (outputs under gpurun_out/<tag>/; copy what is to be judged into profiles/r6/): the
# GPU test suite and the default bench.py line (headline + all legs) the way the driver runs it.  Every command under its own timeout.
set -u
TAG=${1:-r6f}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
tail -3 $OUT/bench_default.time; wc -c $OUT/bench_default.json
