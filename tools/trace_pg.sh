#!/bin/bash
# kernel timeline of ONE LM iteration of the pose-graph path (rocprofv3 kernel trace of bench.py): tools/trace_pg.sh <tag> [bench args]
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_pg_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python $ROOT/bench.py ${@:---steps 4 --warmup 1 --cpu-sample 0 --parity-sample 0} > $OUT/run.log 2>&1)
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void thx::", "").replace("thx::", "").replace("void at::native::", "at::")[:60]
idx = [i for i, r in enumerate(rows) if "pg_assemble_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
print(f"one LM iteration (pg_assemble -> next pg_assemble): {b - a} kernels, {(int(rows[b]['Start_Timestamp']) - t0) / 1e6:.3f} ms")
prev_end, merged = t0, []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = short(r["Kernel_Name"])
    gap = (s - prev_end) / 1e3
    if merged and merged[-1][0] == name and "chol_" in name:
        merged[-1][2] += (e - s) / 1e3; merged[-1][3] += 1; merged[-1][5] = max(merged[-1][5], (e - t0) / 1e3)
    else:
        merged.append([name, (s - t0) / 1e3, (e - s) / 1e3, 1, gap, (e - t0) / 1e3])
    prev_end = max(prev_end, e)
print(f"{'kernel':60s} {'start_us':>10s} {'sum_dur_us':>10s} {'n':>4s} {'gap_before_us':>13s} {'end_us':>10s}")
for m in merged:
    print(f"{m[0]:60s} {m[1]:10.1f} {m[2]:10.1f} {m[3]:4d} {m[4]:13.1f} {m[5]:10.1f}")
PY
find $OUT -name "*.csv" -size +2M -delete
