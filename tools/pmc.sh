#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- <command ...>
# One rocprofv3 --pmc pass per counter group (kernel-trace only, no other trace domain); prints a
# kernel x counter table.  Output under gpurun_out/pmc_<tag>/.
set -u
TAG=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for G in "${GROUPS_[@]}"; do
  (cd /tmp && rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/p$i -o run -- "$@" > $OUT/p$i.log 2>&1)
  i=$((i+1))
done
python $ROOT/tools/summarize_pmc.py $OUT | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +4M -delete
