#!/bin/bash
# round 6: where the right-looking schedule (two launches per block column now) hands over to the left-looking one: batch 36 ... 64 with THX_CHOL_RL_MAX_BATCH = 32 (default) / 64
for rep in 1 2; do
for v in 32 64; do
  export THX_CHOL_RL_MAX_BATCH=$v
  echo "== right_looking_max_batch $v round $rep"
  timeout 300 python tools/batch_sweep.py 36,40,48,56,64 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6
  [ $rep = 1 ] && timeout 300 python tools/ab_small_batch.py 40,48,64 f64 2>&1 | grep "right-looking\|no pairs"
done; done
