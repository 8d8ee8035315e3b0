#!/bin/bash
# A/B: the headline factorisation with the block-compact H gather rounds of the off-diagonal kernels replaced by a register-only
# fake (-DTHX_EXP_NO_GATHER, WRONG results, timing only): the upper bound of what a cheaper gather can buy, fp32 and fp64
O=gpurun_out/${1:-r6w}; mkdir -p $O
for dt in f32 f64; do
  for v in base nogather; do
    if [ $v = base ]; then unset THESEUS_HIP_LIB; else export THESEUS_HIP_LIB=$PWD/theseus_amd/lib/variants/$v.so; fi
    timeout 600 python bench.py --dtype $dt --steps 10 --warmup 3 --legs none --no-sparse-leg --cpu-sample 0 --parity-sample 0 > $O/${dt}_$v.json 2> $O/${dt}_$v.err
    python - $O/${dt}_$v.json $dt $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], sys.argv[3], 'ms/step', round(d['ms_per_step'],3), 'factor frac', round(d['roofline']['frac'],4), 'factor ms', d['roofline'].get('avg_launch_ms'))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'failed', e)
PY
  done
done
