#!/bin/bash
# A/B of two builds of the library on the BA path, one box: kernel trace of tools/bench_ba.py per library.
# usage: tools/ab_ba_lib.sh <tag> <variant.so>
set -u
TAG=$1; VAR=$2
ROOT=$(pwd)
ulimit -c 0
mkdir -p gpurun_out
for lib in "" "$VAR" ""; do
  echo "== lib=${lib:-current} ==" >> gpurun_out/${TAG}_ab_ba_lib.txt
  THESEUS_HIP_LIB=${lib:+$ROOT/$lib} timeout 150 tools/kernel_stats.sh gpurun_out/${TAG}_kstats.txt -- python $ROOT/tools/bench_ba.py > /dev/null
  grep -E "kernel |ba_|chol_" gpurun_out/${TAG}_kstats.txt >> gpurun_out/${TAG}_ab_ba_lib.txt
done
THESEUS_HIP_LIB= timeout 150 python tools/bench_ba.py 2>&1 | tail -25 >> gpurun_out/${TAG}_ab_ba_lib.txt
cat gpurun_out/${TAG}_ab_ba_lib.txt
