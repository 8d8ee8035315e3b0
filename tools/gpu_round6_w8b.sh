#!/bin/bash
# round 6: the 8-wave fp64 off-diagonal kernel for ALL block columns (THX_F64_WIDE_MAX_KTILES=1000) against the 4-wave one (0): other batch sizes, repeats
O=gpurun_out/${1:-r6w8b}; mkdir -p $O
for rep in 1 2; do
for bt in 4096 1024 256; do
for w in 0 1000; do
  export THX_F64_WIDE_MAX_KTILES=$w
  timeout 600 python bench.py --dtype f64 --batch $bt --steps 10 --warmup 3 --legs none --no-sparse-leg --cpu-sample 0 --parity-sample 0 > $O/f64_b${bt}_wide${w}_$rep.json 2> $O/f64_b${bt}_wide${w}_$rep.err
  python - $O/f64_b${bt}_wide${w}_$rep.json $bt $w <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('f64 batch', sys.argv[2], 'wide', sys.argv[3], 'ms/step', round(d['ms_per_step'],3), 'factor frac', round(d['roofline']['frac'],4), 'factor ms', round(d['roofline'].get('avg_launch_ms'),3))
except Exception as e:
    print('f64', sys.argv[2], sys.argv[3], 'failed', e)
PY
done; done; done
