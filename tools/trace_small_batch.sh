#!/bin/bash
# Per-launch timeline of ONE dense factorisation at a small batch (rocprofv3 kernel trace of tools/ab_small_batch.py <B>):
# usage: tools/trace_small_batch.sh <tag> <batch>
set -u
TAG=$1; B=${2:-8}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_small_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python $ROOT/tools/ab_small_batch.py $B > $OUT/run.log 2>&1)
grep -v amdgpu.ids $OUT/run.log | tail -3
python - <<PY
import csv, glob, re
fs = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(chol_\w+_kernel)<([^>]*)>", n)
    return f"{m.group(1)}<{m.group(2)}>" if m else n.split("(")[0].replace("void thx::", "")[:44]
# the LAST complete LM iteration of the run: from the assemble kernel before the last backward-solve launch to the retraction
# after it
lb = max(i for i, r in enumerate(rows) if "chol_bwd" in r["Kernel_Name"])
a = lb
while a > 0 and "assemble" not in rows[a]["Kernel_Name"]:
    a -= 1
last = lb
t0 = int(rows[a]["Start_Timestamp"])
print(f"{'kernel':46s} {'start_us':>9s} {'dur_us':>8s} {'end_us':>9s} {'wgs':>7s}")
end = last
while end + 1 < len(rows) and "retract" not in rows[end]["Kernel_Name"]:
    end += 1
for r in rows[a:end + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wgs = int(r.get("Grid_Size", 0)) // max(int(r.get("Workgroup_Size", 1)), 1)
    print(f"{short(r['Kernel_Name']):46s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(e - t0) / 1e3:9.1f} {wgs:7d}")
PY
find $OUT -name "*.csv" -size +2M -delete
