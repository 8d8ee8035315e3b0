#!/bin/bash
mkdir -p gpurun_out/o
for k in 1 2; do
  for v in base fullinv; do
    if [ $v = base ]; then unset THESEUS_HIP_LIB; else export THESEUS_HIP_LIB=$(pwd)/theseus_amd/lib/variants/$v.so; fi
    echo "== $v" >> gpurun_out/o/ab.txt
    python tools/bench_chol.py 1536 4096 f32 3 2>&1 | grep "fused\|factor " >> gpurun_out/o/ab.txt
  done
done
cat gpurun_out/o/ab.txt
