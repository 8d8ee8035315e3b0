#!/bin/bash
# round 6: the fp32 block-compact SYRK kernel compiled for 4 waves per SIMD (128 VGPRs, no scratch) against 3 (130 VGPRs); and the fp64 suite subset on the new default
O=gpurun_out/${1:-r6syrk32}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_block_hessian.py tests/test_gpu_kernels.py tests/test_gpu_sparse.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
for v in base syrk32_4; do
  if [ $v = base ]; then unset THESEUS_HIP_LIB; else export THESEUS_HIP_LIB=$PWD/theseus_amd/lib/variants/$v.so; fi
  timeout 600 python bench.py --dtype f32 --steps 10 --warmup 3 --legs none --cpu-sample 0 --parity-sample 8 > $O/f32_${v}_$rep.json 2> $O/f32_${v}_$rep.err
  python - $O/f32_${v}_$rep.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('f32', sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'factor frac', round(d['roofline']['frac'],4), 'factor ms', round(d['roofline'].get('avg_launch_ms'),3), 'tile_sparse', round((d.get('tile_sparse') or {}).get('value',0)), 'pose err', (d.get('parity') or {}).get('hip_max_rel_pose_err'))
except Exception as e:
    print('f32', sys.argv[2], 'failed', e)
PY
done; done
