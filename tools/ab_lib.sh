#!/bin/bash
# A/B of two builds of the library on one box: the in-tree libtheseus_hip.so vs a variant .so (THESEUS_HIP_LIB), bench_chol configs.
# usage: tools/ab_lib.sh <tag> <variant.so> "<n B dtype reps>" ...
set -u
TAG=$1; VAR=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do
  for cfg in "$@"; do
    for lib in "" "$VAR"; do
      echo -n "round $round lib=${lib:-current} [$cfg] " >> $OUT/ab_lib.txt
      THESEUS_HIP_LIB=${lib:+$(pwd)/$lib} timeout 300 python tools/bench_chol.py $cfg 2>&1 | grep -E "^n=" | sed 's/; solve.*//' >> $OUT/ab_lib.txt
    done
  done
done
cat $OUT/ab_lib.txt
