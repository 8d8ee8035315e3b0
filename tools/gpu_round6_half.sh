#!/bin/bash
# round 6 experiment: the fp64 HALF-TILE off-diagonal kernel (four workgroups per CU) for the first block columns (THX_F64_HALF_MAX_KTILES)
O=gpurun_out/${1:-r6half}; mkdir -p $O
[ -n "${SKIP_CMP:-}" ] || timeout 600 python tools/cmp_f64_half.py 2>&1 | tail -3
for w in ${HALF_LIST:-0 1 2 3 4 6 0}; do
  export THX_F64_HALF_MAX_KTILES=$w
  timeout 600 python bench.py --dtype f64 --steps 10 --warmup 3 --legs none --no-sparse-leg --cpu-sample 0 --parity-sample 8 > $O/f64_half$w.json 2> $O/f64_half$w.err
  python - $O/f64_half$w.json $w <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('f64 half', sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'factor frac', round(d['roofline']['frac'],4), 'factor ms', round(d['roofline'].get('avg_launch_ms'),3), 'pose err', (d.get('parity') or {}).get('hip_max_rel_pose_err'))
except Exception as e:
    print('f64 half', sys.argv[2], 'failed', e)
PY
done
