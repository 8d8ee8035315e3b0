#!/bin/bash
# late round 4: the bundle-adjustment unrolled backward on the GPU (new kernel thx_ba_unroll_vjp) + the file's other tests
mkdir -p gpurun_out/r4f
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_unrolled.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r4f/pytest_gpu_unrolled.txt 2>&1
echo "exit $?" >> gpurun_out/r4f/pytest_gpu_unrolled.txt
tail -15 gpurun_out/r4f/pytest_gpu_unrolled.txt
