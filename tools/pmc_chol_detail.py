"""Per-dispatch MFMA utilisation of the Cholesky kernels from a tools/pmc.sh run (counter_collection.csv):
util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)."""
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
rows = defaultdict(dict)
names = {}
for f in glob.glob(os.path.join(out, "p*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "chol" not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        rows[k][r["Counter_Name"]] = float(r["Counter_Value"])
        names[k] = r["Kernel_Name"].split("(")[0].replace("void thx::", "").replace("thx::", "")[:28]
for k in sorted(rows)[-26:]:
    c = rows[k]
    if "GRBM_GUI_ACTIVE" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    simd = cyc * 1024
    print(f"{k:5d} {names[k]:28s} cycles/XCD {cyc/1e6:7.2f}M  mfma_busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/simd*100:5.1f}%  "
          f"wave_cyc/slot {c.get('SQ_WAVE_CYCLES',0)*4/ (cyc*256*8)*100:5.1f}%  wait_any {c.get('SQ_WAIT_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1)*100:5.1f}%  "
          f"wait_inst {c.get('SQ_WAIT_INST_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1)*100:5.1f}%  active {c.get('SQ_ACTIVE_INST_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1)*100:5.1f}%  lds_conf {c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',1),1)*100:4.1f}%")
