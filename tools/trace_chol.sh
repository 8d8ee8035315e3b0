#!/bin/bash
# per-launch durations of the Cholesky kernels: tools/trace_chol.sh <tag> [n B dtype]
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python $ROOT/tools/bench_chol.py ${@:-1536 4096 f32 2} > $OUT/run.log 2>&1)
cat $OUT/run.log | grep -v amdgpu.ids
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "chol" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nt = None
out = []
for r in rows:
    name = r["Kernel_Name"].split("<")[0].replace("void thx::", "")
    out.append((name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["VGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"]))
# print the last factor sequence (after warm-up): find last run of diag/offdiag
seq = [o for o in out]
last = len(seq)
print("kernel, us, vgpr, scratch, lds  (last 30 launches)")
for o in seq[-30:]: print("  %-22s %10.1f  %s %s %s" % o)
import collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for o in seq: tot[o[0]] += o[1]; cnt[o[0]] += 1
for k in tot: print(f"{k}: {cnt[k]} launches, total {tot[k]/1e3:.2f} ms")
PY
find $OUT -name "*.csv" -size +2M -delete
