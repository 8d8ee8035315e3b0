#!/bin/bash
# list-driven sparse solves: GPU tests + the sparse / BA benches
mkdir -p gpurun_out/n
timeout 1200 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_ba.py tests/test_gpu_kernels.py tests/test_gpu_implicit.py -x -q -m gpu > gpurun_out/n/pytest.txt 2>&1; tail -5 gpurun_out/n/pytest.txt
python tools/bench_sparse.py 2>&1 | grep -v amdgpu.ids > gpurun_out/n/bench_sparse.txt; tail -3 gpurun_out/n/bench_sparse.txt
python tools/bench_sparse.py 1024 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/n/bench_sparse.txt; tail -3 gpurun_out/n/bench_sparse.txt
python tools/bench_ba.py 2>&1 | grep -v amdgpu.ids > gpurun_out/n/ba_bench_f32.log; tail -3 gpurun_out/n/ba_bench_f32.log
