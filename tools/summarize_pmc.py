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
    if not ("chol" in k or "pg_" in k or "se3_retract" in k or "ba_" in k):
        continue
    print(f"## {k}  ({n} launches) per-launch averages")
    for c, (m, v) in sorted(cs.items()):
        print(f"   {c:34s} {v / m:18.1f}")
    avg = {c: v / m for c, (m, v) in cs.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("GRBM_GUI_ACTIVE"):
        # SQ counters are summed over the whole device (check: SQ_VALU_MFMA_BUSY_CYCLES = 64 cycles x the number of
        # v_mfma_f32_32x32x2 the launch executes); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the launch kept the device
        # busy for GRBM_GUI_ACTIVE / 8 shader cycles on 1024 SIMDs.  Kernels are serialised under counter collection:
        # this is per kernel even where the product path overlaps two streams.
        busy = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8 * 1024)
        print(f"   -> MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = {busy * 100:.1f} %"
              f"   (launch = {avg['GRBM_GUI_ACTIVE'] / 8 / 1e6:.2f} M shader cycles)")
    if "SQ_WAVE_CYCLES" in avg and "SQ_WAIT_ANY" in avg:
        w = avg["SQ_WAVE_CYCLES"]
        print(f"   -> of the wave cycles: parked (s_waitcnt / barrier) {avg['SQ_WAIT_ANY'] / w * 100:.1f} %, issue stall "
              f"{avg.get('SQ_WAIT_INST_ANY', 0) / w * 100:.1f} %, issuing {avg.get('SQ_ACTIVE_INST_ANY', 0) / w * 100:.1f} %")
