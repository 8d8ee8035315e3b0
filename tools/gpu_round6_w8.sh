#!/bin/bash
# round 6: the 8-wave fp64 off-diagonal kernel (chol_offdiag_f64w8_kernel) for the first WIDE block columns of the headline factorisation
O=gpurun_out/${1:-r6w8}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_block_hessian.py -m gpu -x -q -k "eight_wave" 2>&1 | tail -5
for w in 0 1 2 3 4 6 12 0; do
  export THX_F64_WIDE_MAX_KTILES=$w
  timeout 600 python bench.py --dtype f64 --steps 10 --warmup 3 --legs none --no-sparse-leg --cpu-sample 0 --parity-sample 8 > $O/f64_wide$w.json 2> $O/f64_wide$w.err
  python - $O/f64_wide$w.json $w <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('f64 wide', sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'factor frac', round(d['roofline']['frac'],4), 'factor ms', round(d['roofline'].get('avg_launch_ms'),3), 'pose err', (d.get('parity') or {}).get('hip_max_rel_pose_err'))
except Exception as e:
    print('f64 wide', sys.argv[2], 'failed', e)
PY
done
