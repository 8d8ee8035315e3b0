#!/bin/bash
# Device idle time inside the LM iterations of bench.py: kernel trace, then per iteration (pg_assemble -> next pg_assemble)
# the span, the union of the kernel intervals (busy) and their difference.  tools/trace_idle.sh <tag> [bench args]
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_idle_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python $ROOT/bench.py --steps 10 --warmup 2 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none "$@" > $OUT/run.log 2>&1)
grep '"metric"' $OUT/run.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('bench under the profiler: ms_per_step', round(r['ms_per_step'], 3), 'value', round(r['value']))"
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pg_assemble_kernel" in r["Kernel_Name"]][-10:]
tot_span = tot_busy = 0.0
for a, b in zip(idx[:-1], idx[1:]):
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows[a:b])
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    tot_span += span; tot_busy += busy
n = len(idx) - 1
print(f"per LM iteration ({n} iterations, {idx[1] - idx[0]} kernels each): span {tot_span / n / 1e6:.3f} ms, device busy {tot_busy / n / 1e6:.3f} ms, idle {(tot_span - tot_busy) / n / 1e6:.3f} ms = {(1 - tot_busy / tot_span) * 100:.1f} %")
# where the busy time goes: kernel time per iteration by kernel name (sum of durations; launches on two streams overlap)
agg = {}
for r in rows[idx[0]:idx[-1]]:
    k = r["Kernel_Name"].split("(")[0][:90]
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  {t / n / 1e6:9.4f} ms/iter  {c / n:6.1f} launches/iter  {t / c / 1e3:9.2f} us avg  {k}")
PY
find $OUT -name "*.csv" -size +2M -delete
