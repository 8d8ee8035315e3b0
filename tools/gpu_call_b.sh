#!/bin/bash
# gpurun call: GPU tests, bench (constant / adaptive damping), A/B of the Cholesky prologue order.
set -u
TAG=${1:-r2b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -s > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
V=$(pwd)/theseus_amd/lib/variants
for i in 1 2; do
  python tools/bench_chol.py 1536 4096 f32 5 > $OUT/chol_new_$i.txt 2>&1; grep "fused" $OUT/chol_new_$i.txt
  THESEUS_HIP_LIB=$V/prologue_first.so python tools/bench_chol.py 1536 4096 f32 5 > $OUT/chol_old_$i.txt 2>&1; grep "fused" $OUT/chol_old_$i.txt
done
python tools/bench_chol.py 1536 2048 f64 3 > $OUT/chol_f64_new.txt 2>&1; grep "fused" $OUT/chol_f64_new.txt
THESEUS_HIP_LIB=$V/prologue_first.so python tools/bench_chol.py 1536 2048 f64 3 > $OUT/chol_f64_old.txt 2>&1; grep "fused" $OUT/chol_f64_old.txt
python bench.py --cpu-sample 0 --parity-sample 0 > $OUT/bench_f32_const.json 2> $OUT/bench_const.err; python -c "import json;r=json.load(open('$OUT/bench_f32_const.json'));print('const',r['value'],r['ms_per_step'],r['roofline']['frac'])"
python bench.py --adaptive --cpu-sample 0 --parity-sample 0 > $OUT/bench_f32_adaptive.json 2> $OUT/bench_adaptive.err; python -c "import json;r=json.load(open('$OUT/bench_f32_adaptive.json'));print('adaptive',r['value'],r['ms_per_step'],r['roofline']['frac'])"
