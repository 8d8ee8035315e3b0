#!/bin/bash
# bundle adjustment, batch 256: look-ahead schedule (default) vs the half-batch two-stream schedule forced on (THX_CHOL_SPLIT_MIN=256)
mkdir -p gpurun_out/r4u
for round in 1 2; do
  for sm in "" 256; do
    echo "== round $round THX_CHOL_SPLIT_MIN=${sm:-default(1024)}" >> gpurun_out/r4u/ab.txt
    if [ -z "$sm" ]; then unset THX_CHOL_SPLIT_MIN; else export THX_CHOL_SPLIT_MIN=$sm; fi
    timeout 200 python tools/bench_ba.py 512 8192 256 f32 10 2>&1 | grep -E "^optimize|^phases" | cut -c1-260 >> gpurun_out/r4u/ab.txt
  done
done
cat gpurun_out/r4u/ab.txt
