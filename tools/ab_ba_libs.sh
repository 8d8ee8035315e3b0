#!/bin/bash
# A/B of several builds of the library on the BA path, one box: phases of tools/bench_ba.py per library ("" = the in-tree build).
# usage: tools/ab_ba_libs.sh <tag> <variant.so> ...
set -u
TAG=$1; shift
ROOT=$(pwd)
ulimit -c 0
mkdir -p gpurun_out
O=gpurun_out/${TAG}_ab_ba_libs.txt
for round in 1 2; do
  for lib in "" "$@"; do
    echo "== lib=${lib:-current} ==" >> $O
    THESEUS_HIP_LIB=${lib:+$ROOT/$lib} timeout 100 python tools/bench_ba.py 2>&1 | grep -E "phases" | cut -c1-200 >> $O
  done
done
cat $O
