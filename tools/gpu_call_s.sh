#!/bin/bash
# two-stream half-batch schedule at small batches: THX_CHOL_SPLIT_MIN sweep
mkdir -p gpurun_out/s
for sm in 1024 128; do
  echo "== THX_CHOL_SPLIT_MIN=$sm : bundle adjustment batch 256" >> gpurun_out/s/split.txt
  THX_CHOL_SPLIT_MIN=$sm python tools/bench_ba.py 2>&1 | grep "phases\|LM loop" | cut -c1-330 >> gpurun_out/s/split.txt
  for b in 256 512; do
    echo "== THX_CHOL_SPLIT_MIN=$sm : pose graph batch $b" >> gpurun_out/s/split.txt
    THX_CHOL_SPLIT_MIN=$sm python tools/bench_chol.py 1536 $b f32 3 2>&1 | grep "fused" >> gpurun_out/s/split.txt
  done
done
cat gpurun_out/s/split.txt
