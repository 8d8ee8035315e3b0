#!/bin/bash
# bundle adjustment: S as a block list + the level schedule under several camera orders, against the dense-frame path
set -u
mkdir -p gpurun_out/r6g
timeout 900 python -m pytest tests/test_gpu_ba.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r6g/pytest_ba.txt 2>&1; tail -5 gpurun_out/r6g/pytest_ba.txt
for o in natural auto nd1 nd13 md; do
  echo "== ordering $o" | tee -a gpurun_out/r6g/bench_ba.txt
  BENCH_BA_ORDERING=$o timeout 300 python tools/bench_ba.py 512 8192 256 f32 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6g/bench_ba.txt
done
