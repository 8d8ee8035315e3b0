#!/bin/bash
# Per block COLUMN efficiency of the dense factorisation: rocprofv3 kernel trace of tools/bench_chol.py on ONE stream
# (THX_CHOL_SPLIT_MIN=0: the launches of a call run one after the other), the last factor call's launches in order, and for every
# chol_offdiag launch the EXECUTED flops (K-loop 2 j t^3 + substitution 10/16 * 2 t^3 per tile, t = 128) over its duration.
# usage: tools/trace_chol_columns.sh <tag> [n B dtype]         (tools/bench_chol.py: dense H frames)
#        THX_COLS_BENCH=1 tools/trace_chol_columns.sh <tag> n B dtype   (bench.py at that batch / dtype: the LM loop's block-compact H)
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_cols_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ARGS=${@:-1536 4096 f32 2}
set -- $ARGS
if [ -n "${THX_COLS_BENCH:-}" ]; then
  CMD="python $ROOT/bench.py --steps 3 --warmup 1 --batch $2 --dtype $3 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none"
else
  CMD="python $ROOT/tools/bench_chol.py $ARGS"
fi
(cd /tmp && THX_CHOL_SPLIT_MIN=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- $CMD > $OUT/run.log 2>&1)
grep -v amdgpu.ids $OUT/run.log | tail -3
python - $ARGS <<PY
import csv, glob, sys
n, B, dt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
peak = 157.3 if dt == "f32" else 78.6
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f)) if "chol_" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
nt = (n + 127) // 128
# the last complete factor call: nt diagonal phases; walk back from the last bwd kernel
names = [r["Kernel_Name"] for r in rows]
last_off = max(i for i, k in enumerate(names) if "chol_offdiag" in k)
# a call ends with the diagonal phase of the last column (syrk + potrf or diag) after the last offdiag
end = last_off + 1
while end < len(rows) and ("chol_syrk" in names[end] or "chol_potrf" in names[end] or "chol_diag" in names[end]):
    end += 1
# and starts nt - 1 offdiag launches earlier
offs = [i for i, k in enumerate(names[:end]) if "chol_offdiag" in k][-(nt - 1):]
start = offs[0]
# (walk back over column 0's own diagonal phase only: one chol_diag, or chol_syrk + chol_potrf -- what precedes it is the previous call's last column)
start -= 2 if "chol_potrf" in names[start - 1] else 1
t3 = 128.0 ** 3
j = -1
print(f"{'kernel':34s} {'col':>3s} {'wgs':>7s} {'dur_us':>9s} {'exec_TFLOPs':>11s} {'frac':>6s}")
tot = {}
for r in rows[start:end]:
    k = r["Kernel_Name"].split("(")[0].replace("void thx::", "")
    short = k.split("<")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    if "syrk" in short or "chol_diag_kernel" in short:
        j += 1
    fl = None
    rt = wgs // ((B + 7) // 8 * 8)     # row tiles of an off-diagonal launch
    if "offdiag2" in short:            # column pair (j - 1, j): two K-loops over j - 1 tiles, X0 L^T, two substitutions per workgroup
        fl = B * rt * (4 * (j - 1) + 2 + 2.5) * t3
    elif "offdiag" in short:           # one tile per workgroup (column by column; with pairs: the head tile / the last column)
        fl = B * rt * (2 * j * t3 + 1.25 * t3)
    elif "syrk" in short or "chol_diag_kernel" in short:
        fl = B * j * t3 * 36 / 64 * 2 / 2 * 2    # 36 of 64 16x16 blocks of a t x t x (j t) product
    tf = fl / d / 1e6 if fl else float("nan")
    print(f"{short:34s} {j:3d} {wgs:7d} {d:9.1f} {tf:11.1f} {tf / peak:6.3f}")
    tot[short] = tot.get(short, 0.0) + d
print({k: round(v / 1e3, 2) for k, v in tot.items()}, "sum ms", round(sum(tot.values()) / 1e3, 2))
PY
find $OUT -name "*.csv" -size +2M -delete
