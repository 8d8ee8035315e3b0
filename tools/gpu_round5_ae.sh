#!/bin/bash
# Round 5, call ae: round 4's kernels (theseus_amd/lib/variants/r4.so: csrc of commit 9779ed6 + a stub for the new knob) against round 5's
# on ONE box: headline fp32, fp64, implicit batch 1024, the tile-sparse solver, bundle adjustment -- three interleaved rounds.
set -u
TAG=${1:-r5ae}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
COMMON="--cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none"
one() {  # label, lib, args...
  local label=$1 lib=$2; shift 2
  echo -n "$label : " >> $OUT/ab.txt
  THESEUS_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py "$@" 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('value %.0f ms_per_step %.3f factor %.3f frac %.4f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))" >> $OUT/ab.txt
}
for round in 1 2 3; do
  for lib in theseus_amd/lib/variants/r4.so ""; do
    name=${lib:+round-4 kernels}; name=${name:-round-5 kernels}
    one "round $round $name fp32 b4096        " "$lib" --steps 10 --warmup 3 $COMMON
    one "round $round $name fp64 b4096        " "$lib" --steps 6 --warmup 2 $COMMON --dtype f64
    one "round $round $name fp32 b1024        " "$lib" --steps 10 --warmup 3 $COMMON --batch 1024
    one "round $round $name fp32 tile-sparse  " "$lib" --steps 10 --warmup 3 $COMMON --solver sparse
  done
done
cat $OUT/ab.txt
