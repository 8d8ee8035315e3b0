#!/bin/bash
# round 6, H pieces by the matrix cores (hb_scatter): parity tests, then A/B of the headline factorisation, fp32 and fp64,
# THX_HB_SCATTER_MAX_PIECES=0 (the LDS gather rounds, the schedule until now) against the default
O=gpurun_out/${1:-r6x}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_block_hessian.py tests/test_gpu_kernels.py tests/test_gpu_sparse.py tests/test_gpu_ba.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
for dt in f32 f64; do
  for v in lds mfma; do
    if [ $v = lds ]; then export THX_HB_SCATTER_MAX_PIECES=0; else unset THX_HB_SCATTER_MAX_PIECES; fi
    timeout 600 python bench.py --dtype $dt --steps 10 --warmup 3 --legs none --cpu-sample 0 --parity-sample 8 > $O/${dt}_$v.json 2> $O/${dt}_$v.err
    python - $O/${dt}_$v.json $dt $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ts=d.get('tile_sparse') or {}
    print(sys.argv[2], sys.argv[3], 'ms/step', round(d['ms_per_step'],3), 'factor frac', round(d['roofline']['frac'],4), 'factor ms', round(d['roofline'].get('avg_launch_ms'),3), 'value', round(d['value'],1), 'tile_sparse', ts.get('value'), 'parity', {k:v for k,v in (d.get('parity') or {}).items() if 'err' in k})
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'failed', e)
PY
  done
done
