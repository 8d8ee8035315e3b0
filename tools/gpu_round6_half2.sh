#!/bin/bash
# round 6: the fp64 half-tile kernel at other batch sizes / n (THX_F64_HALF_MAX_KTILES 0 vs 8), and the test
O=gpurun_out/${1:-r6half3}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_block_hessian.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
for cfg in "4096 256" "1024 256" "256 256" "256 512" "64 256"; do
  set -- $cfg
  for w in 0 8; do
    export THX_F64_HALF_MAX_KTILES=$w
    timeout 600 python bench.py --dtype f64 --batch $1 --poses $2 --edges $((4*$2)) --steps 10 --warmup 3 --legs none --no-sparse-leg --cpu-sample 0 --parity-sample 0 > $O/f64_b$1_p$2_half$w.json 2> $O/f64_b$1_p$2_half$w.err
    python - $O/f64_b$1_p$2_half$w.json $1 $2 $w <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('f64 batch', sys.argv[2], 'poses', sys.argv[3], 'half', sys.argv[4], 'ms/step', round(d['ms_per_step'],3), 'factor frac', round(d['roofline']['frac'],4), 'factor ms', round(d['roofline'].get('avg_launch_ms'),3))
except Exception as e:
    print('f64', sys.argv[2], sys.argv[3], sys.argv[4], 'failed', e)
PY
  done
done
