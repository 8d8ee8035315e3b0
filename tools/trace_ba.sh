#!/bin/bash
# kernel timeline of ONE LM iteration of the bundle-adjustment path (rocprofv3 kernel trace of tools/bench_ba.py):
# tools/trace_ba.sh <tag> [bench_ba args]
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_ba_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python $ROOT/tools/bench_ba.py ${@:-512 8192 256 f32 5} > $OUT/run.log 2>&1)
grep -v amdgpu.ids $OUT/run.log | tail -4
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void thx::", "").replace("thx::", "").replace("void at::native::", "at::")[:60]
# the last optimize(): iterations start at ba_point_kernel launches; take the last-but-one complete iteration
idx = [i for i, r in enumerate(rows) if "ba_point_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
print(f"one LM iteration: {b - a} kernels, {(int(rows[b]['Start_Timestamp']) - t0) / 1e6:.2f} ms")
prev_end, merged = t0, []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = short(r["Kernel_Name"])
    gap = (s - prev_end) / 1e3
    if merged and merged[-1][0] == name and gap < 50:
        merged[-1][2] += (e - s) / 1e3; merged[-1][3] += 1; merged[-1][4] += max(gap, 0)
    else:
        merged.append([name, (s - t0) / 1e3, (e - s) / 1e3, 1, max(gap, 0)])
    prev_end = max(prev_end, e)
# idle gaps of the device over the whole last optimize() (its iterations = the last ITERS ba_point_kernel launches)
iters = int("${5:-5}") if "${5:-5}".isdigit() else 5
lo = idx[-iters] if len(idx) >= iters else idx[0]
pe = int(rows[lo]["Start_Timestamp"])
busy = 0
print("device idle gaps > 300 us from the first iteration of the last optimize() to the end of the trace:")
for r in rows[lo:]:
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s_ - pe > 300e3:
        print(f"   {(s_ - pe) / 1e6:8.2f} ms idle before {short(r['Kernel_Name'])} at +{(s_ - int(rows[lo]['Start_Timestamp'])) / 1e6:.2f} ms")
    busy += e_ - max(s_, pe) if e_ > pe else 0
    pe = max(pe, e_)
print(f"   span {(pe - int(rows[lo]['Start_Timestamp'])) / 1e6:.2f} ms, device busy {busy / 1e6:.2f} ms")
print(f"{'kernel':60s} {'start_us':>10s} {'busy_us':>10s} {'n':>5s} {'gap_before_us':>13s}")
for m in merged:
    print(f"{m[0]:60s} {m[1]:10.1f} {m[2]:10.1f} {m[3]:5d} {m[4]:13.1f}")
PY
find $OUT -name "*.csv" -size +2M -delete
