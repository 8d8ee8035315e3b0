#!/bin/bash
# round 6: the right-looking schedule's two launch arrangements (THX_CHOL_RL_LOOKAHEAD 1 / 2) at the upper end of its batch range, fp32
for rep in 1 2; do
for v in 1 2; do
  export THX_CHOL_RL_LOOKAHEAD=$v
  echo "== fp32 mode $v round $rep"
  timeout 300 python tools/batch_sweep.py 24,32,40,48,56,64 2>&1 | grep -v "^$\|amdgpu.ids" | tail -7
done; done
