#!/bin/bash
# round 6 experiment: the two-stream schedule of the dense factorisation with the second stream LAG block columns behind the first
# and the batch in 2 x PIPE parts (THX_CHOL_LAG_COLS / THX_CHOL_PIPE): do one stream's short-K columns fill under the other's long ones?
O=gpurun_out/${1:-r6lag}; mkdir -p $O
run() {  # dtype lag pipe
  if [ $2 = 0 ]; then unset THX_CHOL_LAG_COLS THX_CHOL_PIPE; else export THX_CHOL_LAG_COLS=$2 THX_CHOL_PIPE=$3; fi
  timeout 600 python bench.py --dtype $1 --steps 10 --warmup 3 --legs none --no-sparse-leg --cpu-sample 0 --parity-sample 8 > $O/$1_lag$2_pipe$3.json 2> $O/$1_lag$2_pipe$3.err
  python - $O/$1_lag$2_pipe$3.json $1 $2 $3 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'lag', sys.argv[3], 'pipe', sys.argv[4], 'ms/step', round(d['ms_per_step'],3), 'factor frac', round(d['roofline']['frac'],4), 'factor ms', round(d['roofline'].get('avg_launch_ms'),3), 'pose err', (d.get('parity') or {}).get('hip_max_rel_pose_err'))
except Exception as e:
    print(sys.argv[2], sys.argv[3], sys.argv[4], 'failed', e)
PY
}
for dt in f64 f32; do
  run $dt 0 0
  for lag in 4 6 8; do for pipe in 1 2; do run $dt $lag $pipe; done; done
  run $dt 6 4
  run $dt 0 0
done
