#!/bin/bash
python -m pytest tests/test_gpu_ba.py -q -m gpu -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bap -o run -- python /root/repo/tools/bench_ba.py > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/bap/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("lm_accept", "ba_", "vec_retract")):
        print(r["Name"].split("(")[0][:60].ljust(62), r["Calls"].rjust(4), "%9.1f us avg" % (float(r["AverageNs"])/1e3))
PY
