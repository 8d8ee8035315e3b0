#!/bin/bash
set -u
mkdir -p gpurun_out/r6v
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "chol" > gpurun_out/r6v/pytest_chol.txt 2>&1; tail -6 gpurun_out/r6v/pytest_chol.txt
timeout 600 python tools/ab_small_batch.py 8,16,32,64 f64 2>&1 | grep -v "amdgpu.ids\|no pairs" | tee gpurun_out/r6v/ab_small_batch_f64.txt
