#!/bin/bash
# A/B of the XCD-aware block order of ba_schur_block_kernel (THX_BA_XCD_ORDER=0|1) on one box: per-kernel time from the
# kernel trace of tools/bench_ba.py.   tools/ab_ba_xcd.sh <tag> [orders, default "0 1"]
set -u
TAG=${1:-x}
ORDERS=${2:-"0 1"}
ROOT=$(pwd)
ulimit -c 0
mkdir -p gpurun_out
for X in $ORDERS; do
  export THX_BA_XCD_ORDER=$X
  echo "== THX_BA_XCD_ORDER=$X ==" >> gpurun_out/${TAG}_ab_ba_xcd.txt
  timeout 150 tools/kernel_stats.sh gpurun_out/${TAG}_ba_kstats_$X.txt -- python $ROOT/tools/bench_ba.py > /dev/null
  grep -E "kernel |ba_|chol_" gpurun_out/${TAG}_ba_kstats_$X.txt >> gpurun_out/${TAG}_ab_ba_xcd.txt
done
cat gpurun_out/${TAG}_ab_ba_xcd.txt
