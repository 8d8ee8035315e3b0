#!/bin/bash
# One gpurun call: GPU test suite, headline bench (fp32 / fp64), in-kernel stamps of the Cholesky kernels, the drop-in.
set -u
TAG=${1:-r2a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -s > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err; tail -c 1500 $OUT/bench_f32.json
python bench.py --dtype f64 --steps 5 --cpu-sample 32 > $OUT/bench_f64.json 2> $OUT/bench_f64.err; tail -c 900 $OUT/bench_f64.json
V=$(pwd)/theseus_amd/lib/variants
THESEUS_HIP_LIB=$V/offprof.so python tools/prof/off_prof.py > $OUT/off_prof.txt 2>&1; cat $OUT/off_prof.txt
THESEUS_HIP_LIB=$V/diagprof.so python tools/prof/diag_prof.py > $OUT/diag_prof.txt 2>&1; tail -40 $OUT/diag_prof.txt
[ -d _refcopy ] && tools/dropin_gpu.sh
