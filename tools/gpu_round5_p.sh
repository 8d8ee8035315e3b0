#!/bin/bash
# Round 5, call p: WHERE the off-diagonal launch loses its time -- per-workgroup residency (tools/prof/off_occupancy.py) and
# several tiles per workgroup (THX_OFF_TPW) against one.
set -u
TAG=${1:-r5p}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
for tpw in 1 12; do
  echo "== THX_OFF_TPW=$tpw, one stream" >> $OUT/occupancy.txt
  THX_OFF_TPW=$tpw THX_CHOL_SPLIT_MIN=0 THESEUS_HIP_LIB=$PWD/theseus_amd/lib/variants/prof32.so timeout 300 python tools/prof/off_occupancy.py 4096 2>&1 | grep -v amdgpu.ids >> $OUT/occupancy.txt
done
cat $OUT/occupancy.txt
for round in 1 2; do
  for split in 0 1024; do
    for tpw in 1 2 3 4 6 12; do
      echo -n "round $round split_min=$split tpw=$tpw : " >> $OUT/ab.txt
      THX_OFF_TPW=$tpw THX_CHOL_SPLIT_MIN=$split timeout 300 python tools/bench_chol.py 1536 4096 f32 5 2>&1 | grep -E "^n=" | sed 's/; solve.*//' >> $OUT/ab.txt
    done
  done
done
cat $OUT/ab.txt
THX_OFF_TPW=3 timeout 900 python -m pytest tests/test_gpu_block_hessian.py tests/test_gpu_kernels.py tests/test_gpu_sparse.py -q -x -p no:cacheprovider > $OUT/pytest_tpw3.txt 2>&1; tail -3 $OUT/pytest_tpw3.txt
