#!/bin/bash
# where the block-compact H costs: the headline factorisation (fp32, fp64) with (a) the product, (b) no table / value loads
# (-DTHX_EXP_NO_HBLOAD, fake dense values, WRONG results), (c) no loads and no barrier around the list
O=gpurun_out/${1:-r6y}; mkdir -p $O
for dt in f32 f64; do
  for v in base nohbload base nohbload; do
    if [ $v = base ]; then unset THESEUS_HIP_LIB; else export THESEUS_HIP_LIB=$PWD/theseus_amd/lib/variants/$v.so; fi
    timeout 600 python bench.py --dtype $dt --steps 10 --warmup 3 --legs none --no-sparse-leg --cpu-sample 0 --parity-sample 0 > $O/${dt}_$v.json 2> $O/${dt}_$v.err
    python - $O/${dt}_$v.json $dt $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], sys.argv[3], 'ms/step', round(d['ms_per_step'],3), 'factor frac', round(d['roofline']['frac'],4), 'factor ms', round(d['roofline'].get('avg_launch_ms'),3))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'failed', e)
PY
  done
done
