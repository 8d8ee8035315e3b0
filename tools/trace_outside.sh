#!/bin/bash
# What does one optimize() run on the device OUTSIDE its LM iterations?  Kernel trace of bench.py; the timed forward = the last
# K pg_assemble launches; kernels between the previous forward's last chol_bwd and the first pg_assemble (prologue) and after
# the last iteration (epilogue), grouped by name.  tools/trace_outside.sh <tag> [bench args]
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_outside_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
K=5
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python $ROOT/bench.py --steps $K --warmup 2 --cpu-sample 0 --parity-sample 0 --no-sparse-leg "$@" > $OUT/run.log 2>&1)
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void thx::", "").replace("thx::", "").replace("void at::native::", "at::")[:80]
asm = [i for i, r in enumerate(rows) if "pg_assemble_kernel" in r["Kernel_Name"]]
bwd = [i for i, r in enumerate(rows) if "chol_bwd_kernel" in r["Kernel_Name"]]
first = asm[-$K]
prev_bwd = max(i for i in bwd if i < first)
last_bwd = bwd[-1]
def report(tag, lo, hi):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[lo:hi]:
        k = short(r["Kernel_Name"]); d[k][0] += 1; d[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    span = (int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6 if hi > lo else 0.0
    print(f"{tag}: {hi - lo} kernels, span {span:.3f} ms, busy {sum(v[1] for v in d.values()):.3f} ms")
    for k, (n, ms) in sorted(d.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"   {ms:8.3f} ms  x{n:<5d} {k}")
report("between the previous forward's last solve and this forward's first assemble (epilogue of one + prologue of the next)", prev_bwd + 1, first)
report("after the last solve of the timed forward", last_bwd + 1, len(rows))
PY
find $OUT -name "*.csv" -size +2M -delete
