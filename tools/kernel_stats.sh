#!/bin/bash
# rocprofv3 --kernel-trace --stats of any command, top kernels as a table: tools/kernel_stats.sh <out.txt> -- <command ...>
set -u
OUTF=$1; shift; shift
ROOT=$(pwd); D=/tmp/kstats_$$; mkdir -p $D
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $D -o run -- "$@" > $D/run.log 2>&1)
python - <<PY > $ROOT/$OUTF
import csv, glob
f = glob.glob("$D/**/*kernel_stats.csv", recursive=True)[0]
print("# rocprofv3 --kernel-trace --stats -- $*")
print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
for r in list(csv.DictReader(open(f)))[:22]:
    print(f"{r['Name'].split('(')[0].replace('void thx::','').replace('thx::','')[:70]:72s} {r['Calls']:>6s} {int(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
PY
rm -rf $D
cat $ROOT/$OUTF | head -14
