#!/bin/bash
# round 6: the fp64 block-compact SYRK kernel compiled for 3 / 4 waves per SIMD (168 VGPRs + 20 B scratch / 128 VGPRs + 168 B scratch) against 2 (200 VGPRs)
O=gpurun_out/${1:-r6syrk}; mkdir -p $O
for rep in 1 2; do
for v in base syrk3 syrk4; do
  if [ $v = base ]; then unset THESEUS_HIP_LIB; else export THESEUS_HIP_LIB=$PWD/theseus_amd/lib/variants/$v.so; fi
  timeout 600 python bench.py --dtype f64 --steps 10 --warmup 3 --legs none --no-sparse-leg --cpu-sample 0 --parity-sample 8 > $O/f64_${v}_$rep.json 2> $O/f64_${v}_$rep.err
  python - $O/f64_${v}_$rep.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('f64', sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'factor frac', round(d['roofline']['frac'],4), 'factor ms', round(d['roofline'].get('avg_launch_ms'),3), 'pose err', (d.get('parity') or {}).get('hip_max_rel_pose_err'))
except Exception as e:
    print('f64', sys.argv[2], 'failed', e)
PY
done; done
