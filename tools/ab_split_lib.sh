#!/bin/bash
# current build vs an older build (variant .so), split vs fused diagonal phase, bench_chol at the headline size, one box
set -u
TAG=$1; VAR=$2; OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do
  for lib in "" "$VAR"; do
    for smin in 2048 1000000; do
      echo -n "round $round lib=${lib:-current} split_diag_min=$smin " >> $OUT/ab_split_lib.txt
      THX_CHOL_SPLIT_DIAG_MIN=$smin THESEUS_HIP_LIB=${lib:+$(pwd)/$lib} timeout 300 python tools/bench_chol.py 1536 4096 f32 3 2>&1 | grep -E "^n=" | sed 's/; solve.*//' >> $OUT/ab_split_lib.txt
    done
  done
done
cat $OUT/ab_split_lib.txt
