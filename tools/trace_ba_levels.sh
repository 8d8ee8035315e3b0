#!/bin/bash
# Per-launch timeline of ONE linear solve of the bundle-adjustment leg (rocprofv3 kernel trace of tools/bench_ba.py):
# usage: tools/trace_ba_levels.sh <tag> [ordering]
set -u
TAG=$1; ORD=${2:-auto}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_ba_$TAG; mkdir -p $OUT
export TMPDIR=/tmp BENCH_BA_ORDERING=$ORD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python $ROOT/tools/bench_ba.py 512 8192 256 f32 3 > $OUT/run.log 2>&1)
grep -v amdgpu.ids $OUT/run.log | tail -3
python - <<PY
import csv, glob, re
fs = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)
if not fs:
    raise SystemExit("no kernel trace csv under $OUT")
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(chol_\w+_kernel)<([^>]*)>", n)
    return f"{m.group(1)}<{m.group(2)}>" if m else n.split("(")[0].replace("void thx::", "")[:44]
# the LAST linear solve: from the last ba_point_invert kernel to the next ba_backsub
inv = [i for i, r in enumerate(rows) if "ba_point_invert" in r["Kernel_Name"]]
a = inv[-2] if len(inv) > 1 and "$TAG".endswith("cycle") else inv[-1]   # (<tag>cycle: from the solve before the last one on)
t0 = int(rows[a]["Start_Timestamp"])
print(f"{'kernel':46s} {'start_us':>9s} {'dur_us':>8s} {'end_us':>9s} {'wgs':>7s}")
tot = {}
for r in rows[a:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wgs = int(r.get("Grid_Size", 0)) // max(int(r.get("Workgroup_Size", 1)), 1)
    print(f"{short(r['Kernel_Name']):46s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(e - t0) / 1e3:9.1f} {wgs:7d}")
    k = short(r['Kernel_Name']).split("<")[0]
    tot[k] = tot.get(k, 0) + (e - s) / 1e3
    if "ba_backsub" in r["Kernel_Name"] and not ("$TAG".endswith("cycle") and rows.index(r) < inv[-1]):
        break
print("busy per kernel (us):", {k: round(v, 1) for k, v in tot.items()})
PY
find $OUT -name "*.csv" -size +2M -delete
