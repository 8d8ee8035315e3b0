#!/bin/bash
# round 6: the look-ahead schedule (HEAD tile + the next diagonal phase on one stream, the REST of the column on a second) on DENSE frames at
# the batch sizes whose launches do not fill the chip (THX_CHOL_LOOKAHEAD_DENSE_MAX_BATCH): 256 poses, batch 48 ... 512
O=gpurun_out/${1:-r6la}; mkdir -p $O
for rep in 1 2; do
for v in 0 100000; do
  export THX_CHOL_LOOKAHEAD_DENSE_MAX_BATCH=$v
  echo "== dense look-ahead max batch $v round $rep"
  timeout 300 python tools/batch_sweep.py 48,64,96,128,192,256,512 2>&1 | grep -v "^$" | tail -8
  if [ $rep = 1 ]; then
    for b in 64 128; do timeout 300 python tools/bench_chol.py 1536 $b f32 2>&1 | tail -3; done
    timeout 300 python tools/bench_chol.py 1536 64 f64 2>&1 | tail -3
    timeout 300 python tools/bench_chol.py 3072 64 f32 2>&1 | tail -3
  fi
done; done
