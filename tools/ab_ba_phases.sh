#!/bin/bash
# same-box A/B of a variant library on the bundle-adjustment path: tools/bench_ba.py phases (event-timed), current / variant, twice
# usage: tools/ab_ba_phases.sh <tag> <variant.so>
set -u
TAG=$1; VAR=$2; OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do
  for lib in "" "$VAR"; do
    echo "== round $round lib=${lib:-current}" >> $OUT/ab.txt
    THESEUS_HIP_LIB=${lib:+$(pwd)/$lib} timeout 200 python tools/bench_ba.py 512 8192 256 f32 10 2>&1 | grep -E "^optimize|^phases|tile-sparse" | cut -c1-330 >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
