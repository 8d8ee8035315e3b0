"""VGPRs / scratch / occupancy of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
usage: python tools/kernel_resources.py theseus_amd/csrc/chol_kernels.hip [extra hipcc flags]"""
import re
import subprocess
import sys

cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed",
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[2:], "-c", sys.argv[1], "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (?:\S+:\d+:\d+: )?\s*(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif ":" in t and cur is not None:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n.replace("void thx::", ""))
    g = lambda k: r.get(k, "?")  # noqa: E731
    print(f"{n:58s} VGPRs {g('VGPRs'):>4s} AGPRs {g('AGPRs'):>4s} scratch {g('ScratchSize [bytes/lane]'):>5s} "
          f"occupancy {g('Occupancy [waves/SIMD]')} LDS {g('LDS Size [bytes/block]')}")
