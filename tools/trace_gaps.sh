#!/bin/bash
# Where does the time of one optimize() go BETWEEN kernels?  rocprofv3 kernel trace of bench.py, then every gap > 0.3 ms
# between consecutive kernels of the last (timed) forward, with the kernels around it.  tools/trace_gaps.sh <tag> [bench args]
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_gaps_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python $ROOT/bench.py ${@:---steps 10 --warmup 2 --cpu-sample 0 --parity-sample 0} > $OUT/run.log 2>&1)
grep '"metric"' $OUT/run.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('ms_per_step', r['ms_per_step'], 'value', r['value'])"
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void thx::", "").replace("thx::", "").replace("void at::native::", "at::")[:70]
idx = [i for i, r in enumerate(rows) if "pg_assemble_kernel" in r["Kernel_Name"]]
# the timed forward = the last 10 pg_assemble launches; start a little before the first of them
a = idx[-10]
lo = max(0, a - 400)
t0 = int(rows[a]["Start_Timestamp"])
prev_end, prev_name = None, None
print(f"kernels {len(rows)}, timed forward from kernel {a}; span first assemble -> last kernel: {(int(rows[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms")
for i in range(lo, len(rows)):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and s - prev_end > 300000:
        print(f"gap {(s - prev_end) / 1e6:8.3f} ms at t = {(s - t0) / 1e6:9.3f} ms : after [{prev_name}]  before [{short(r['Kernel_Name'])}]")
    if prev_end is None or e > prev_end:
        prev_end, prev_name = e, short(r["Kernel_Name"])
# how long before the first assemble did the forward's first kernel start?
big = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows[lo:a]]
print(f"400 kernels before the first assemble: span {(t0 - int(rows[lo]['Start_Timestamp'])) / 1e6:.2f} ms, busy {sum(big):.2f} ms")
PY
find $OUT -name "*.csv" -size +2M -delete
