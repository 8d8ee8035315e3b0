#!/bin/bash
# fp64 A/B: chol_potrf_kernel<double> at 512 registers (one wave per SIMD: needs an EMPTY SIMD, starved by the other stream's
# chol_offdiag) vs a variant library (default: bounded to 256 registers, the rest spilled).  usage: tools/ab_potrf64.sh tag [variant.so]
set -u
TAG=${1:-abp}; VAR=${2:-theseus_amd/lib/variants/potrf64occ2.so}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do
  for lib in "" "$VAR"; do
    echo "== round $round lib=${lib:-current}" >> $OUT/ab.txt
    THESEUS_HIP_LIB=${lib:+$(pwd)/$lib} timeout 300 python tools/bench_chol.py 1536 4096 f64 2 2>&1 | grep -v amdgpu | grep -E "^n=|^fused|resid" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
