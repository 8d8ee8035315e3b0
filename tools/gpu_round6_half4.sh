#!/bin/bash
# round 6: the fp64 half-tile kernel's solve panel through a ring of four slots (product) against the three-phase version (-DTHX_F64H_PHASES)
O=gpurun_out/${1:-r6half5}; mkdir -p $O
timeout 600 python tools/cmp_f64_half.py 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_block_hessian.py -m gpu -x -q -k "eight_wave or half" 2>&1 | tail -2
for rep in 1 2; do
for v in base f64h_phases; do
  if [ $v = base ]; then unset THESEUS_HIP_LIB; else export THESEUS_HIP_LIB=$PWD/theseus_amd/lib/variants/$v.so; fi
  for w in 8 12; do
    export THX_F64_HALF_MAX_KTILES=$w
    timeout 600 python bench.py --dtype f64 --steps 10 --warmup 3 --legs none --no-sparse-leg --cpu-sample 0 --parity-sample 8 > $O/f64_${v}_half${w}_$rep.json 2> $O/f64_${v}_half${w}_$rep.err
    python - $O/f64_${v}_half${w}_$rep.json $v $w <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('f64', sys.argv[2], 'half', sys.argv[3], 'ms/step', round(d['ms_per_step'],3), 'factor frac', round(d['roofline']['frac'],4), 'factor ms', round(d['roofline'].get('avg_launch_ms'),3), 'pose err', (d.get('parity') or {}).get('hip_max_rel_pose_err'))
except Exception as e:
    print('f64', sys.argv[2], sys.argv[3], 'failed', e)
PY
  done
done; done
