#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace + stats of bench.py, then the HBM counters in their own
# passes (never mixed with other trace domains), summaries copied to gpurun_out/prof_<tag>/.
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ARGS=${@:-"--steps 3 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none"}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o bench -- python $ROOT/bench.py $ARGS > $OUT/bench_pmc_$C.log 2>&1
done
cd $ROOT
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merged directory small: drop the raw per-dispatch traces (summaries stay)
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
