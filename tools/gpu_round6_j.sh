#!/bin/bash
set -u
mkdir -p gpurun_out/r6j
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "chol" > gpurun_out/r6j/pytest_chol.txt 2>&1; tail -15 gpurun_out/r6j/pytest_chol.txt
timeout 600 python tools/ab_small_batch.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6j/ab_small_batch.txt
