#!/bin/bash
set -u
mkdir -p gpurun_out/r6k
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "chol" > gpurun_out/r6k/pytest_chol.txt 2>&1; tail -8 gpurun_out/r6k/pytest_chol.txt
timeout 600 python tools/ab_small_batch.py 8,16,32,64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6k/ab_small_batch.txt
bash tools/trace_small_batch.sh b8 8 > gpurun_out/r6k/trace_small_b8.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=30 > gpurun_out/r6k/pytest_gpu.txt 2>&1; tail -15 gpurun_out/r6k/pytest_gpu.txt
