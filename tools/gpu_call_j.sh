#!/bin/bash
# K-loop microbenchmarks + the factorisation (dense and tile-sparse) on the same box
mkdir -p gpurun_out/j
theseus_amd/lib/variants/kloop_pieces > gpurun_out/j/kloop_pieces.txt 2>&1
python tools/bench_chol.py 1536 4096 f32 3 > gpurun_out/j/bench_chol.txt 2>&1
python tools/bench_sparse.py > gpurun_out/j/bench_sparse.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/j/pytest.txt 2>&1
cut -c1-150 gpurun_out/j/kloop_pieces.txt; tail -4 gpurun_out/j/bench_chol.txt; tail -12 gpurun_out/j/bench_sparse.txt; tail -3 gpurun_out/j/pytest.txt
