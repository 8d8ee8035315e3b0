#!/bin/bash
# bundle adjustment, batch 256: fused diagonal phase (default below THX_CHOL_SPLIT_DIAG_MIN) vs SYRK + potrf kernels
mkdir -p gpurun_out/r4v
for round in 1 2; do
  for sm in "" 256; do
    echo "== round $round THX_CHOL_SPLIT_DIAG_MIN=${sm:-default}" >> gpurun_out/r4v/ab.txt
    if [ -z "$sm" ]; then unset THX_CHOL_SPLIT_DIAG_MIN; else export THX_CHOL_SPLIT_DIAG_MIN=$sm; fi
    timeout 200 python tools/bench_ba.py 512 8192 256 f32 10 2>&1 | grep -E "^optimize|^phases" | cut -c1-260 >> gpurun_out/r4v/ab.txt
  done
done
cat gpurun_out/r4v/ab.txt
