#!/bin/bash
# Per-launch timeline of ONE tile-sparse factorisation of the bundle-adjustment reduced system (rocprofv3 kernel trace of
# tools/bench_ba.py): every chol_* launch between a ba_schur_kernel and the following chol_bwd_kernel, with its stream-relative
# start, duration and template arguments (MODE 1 / 2 = the early / late launches of the deep look-ahead schedule).
# usage: [THX_CHOL_DEEP=0|1 ...] tools/trace_chol_sparse.sh <tag>
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_chol_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python $ROOT/tools/bench_ba.py 512 8192 256 f32 5 > $OUT/run.log 2>&1)
grep -v amdgpu.ids $OUT/run.log | grep -E "phases" | tail -1
python - <<PY
import csv, glob, re
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(chol_\w+_kernel)<([^>]*)>", n)
    return f"{m.group(1)}<{m.group(2)}>" if m else n.split("(")[0].replace("void thx::", "")[:40]
# the last LM iteration's factorisation: from the last ba_schur_kernel to the next chol_bwd_kernel
a = max(i for i, r in enumerate(rows) if "ba_schur_kernel" in r["Kernel_Name"])
b = next(i for i in range(a, len(rows)) if "chol_bwd_kernel" in rows[i]["Kernel_Name"])
t0 = int(rows[a]["End_Timestamp"])
print(f"{'kernel':44s} {'queue':>6s} {'start_us':>9s} {'dur_us':>8s} {'end_us':>9s} {'wgs':>6s}")
for r in rows[a + 1:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{short(r['Kernel_Name']):44s} {r.get('Queue_Id', '?'):>6s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(e - t0) / 1e3:9.1f} "
          f"{int(r.get('Grid_Size', r.get('Grid_Size_X', 0))) // max(int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 1))), 1):6d}")
print(f"span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
PY
find $OUT -name "*.csv" -size +2M -delete
