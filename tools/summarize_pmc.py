"""kernel x counter table from rocprofv3 --pmc passes (tools/pmc.sh): per-launch averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
tab = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(out, "p*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void thx::", "").replace("thx::", "")[:48]
        a = tab[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, cs in sorted(tab.items()):
    n = max(v[0] for v in cs.values())
    if not ("chol" in k or "pg_" in k or "se3_retract" in k):
        continue
    print(f"## {k}  ({n} launches) per-launch averages")
    for c, (m, v) in sorted(cs.items()):
        print(f"   {c:34s} {v / m:18.1f}")
