#!/bin/bash
# round 6: the 32 x 32 diagonal sub-block factorised + inverted in ONE pass over 64 lanes (potrf_inv32_lanes: the identity's rows in lanes 32..63 take the
# factorisation's column operations) against the blocked 16 + 16 scheme with inv_tri (-DTHX_POTRF_BLOCKED = variants/blocked.so)
O=gpurun_out/${1:-r6potrf}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sparse.py tests/test_gpu_block_hessian.py -m gpu -x -q 2>&1 | tail -15
for rep in 1 2; do
for v in base blocked; do
  if [ $v = base ]; then unset THESEUS_HIP_LIB; else export THESEUS_HIP_LIB=$PWD/theseus_amd/lib/variants/$v.so; fi
  echo "== $v round $rep"
  timeout 300 python tools/batch_sweep.py 8,16,32,64,256 2>&1 | grep -v "^$" | tail -8
  timeout 300 python tools/bench_sparse.py 4096 64 f32 10 2>&1 | tail -4
  timeout 300 python tools/bench_sparse.py 4096 256 f32 10 2>&1 | tail -4
  timeout 300 python tools/bench_ba.py 2>&1 | grep "phases\|per solve" | tail -3
  if [ $rep = 1 ]; then
  for dt in f32 f64; do
  timeout 600 python bench.py --dtype $dt --steps 10 --warmup 3 --legs none --no-sparse-leg --cpu-sample 0 --parity-sample 8 > $O/${dt}_${v}.json 2> $O/${dt}_${v}.err
  python - $O/${dt}_${v}.json $v $dt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[3], sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'factor frac', round(d['roofline']['frac'],4), 'factor ms', round(d['roofline'].get('avg_launch_ms'),3), 'pose err', (d.get('parity') or {}).get('hip_max_rel_pose_err'))
except Exception as e:
    print(sys.argv[3], sys.argv[2], 'failed', e)
PY
  done
  fi
done; done
