"""Where and when the workgroups of ONE chol_offdiag_f32 launch ran (library built with -DTHX_OFF_PROF, run with
THESEUS_HIP_LIB=<that .so> and THX_CHOL_SPLIT_MIN=0: one stream, nothing overlapped): every workgroup leaves HW_ID, XCC_ID and
its first / last wall-clock tick (100 MHz) in the head of its output tile.  Per block column: launch span, workgroup lifetime,
residency per CU over the span (fraction of the time with 0 / 1 / 2 workgroups), gap between consecutive workgroups of a slot.
usage: python tools/prof/off_occupancy.py [batch] [columns ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from theseus_amd.kernels import default_kernels, round_up

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
COLS = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 5, 8, 10]
n, dt = 1536, torch.float32
K = default_kernels(); ld = round_up(n, 32)
nt = n // 128

gen = torch.Generator(device="cuda").manual_seed(0)
H = torch.empty(B, ld, ld, dtype=dt, device="cuda"); H.uniform_(-1, 1, generator=gen)
H.diagonal(dim1=1, dim2=2).add_(float(n))
L = torch.zeros_like(H); P = torch.empty(B, nt, 128, 128, dtype=dt, device="cuda")
info = torch.empty(B, dtype=torch.int32, device="cuda"); lam = torch.full((B,), 1e-3, dtype=dt, device="cuda")
for _ in range(2):
    K.chol_factor(H, n, lam, False, 1e-8, L, P, info)
torch.cuda.synchronize()

for j in COLS:
    rows = []
    for i in range(j + 1, nt):
        head = L[:, 128 * i, 128 * j:128 * j + 12].contiguous()
        cyc = head[:, 1:7].double().cpu().numpy()
        raw = head.view(torch.int32)[:, 7:11].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        rows.append(np.concatenate([cyc, raw], axis=1))
    a = np.concatenate(rows, 0)
    hw, xcc, t0, t1 = a[:, 6].astype(np.int64), a[:, 7].astype(np.int64) & 0xF, a[:, 8], a[:, 9]
    t1 = np.where(t1 < t0, t1 + 2.0 ** 32, t1)
    base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0     # us
    cu = (xcc << 8) | ((hw >> 8) & 0xFF)      # XCC, SE / SH / CU bits of HW_ID
    span = t1.max()
    life = t1 - t0
    ncu = len(np.unique(cu))
    # residency: sweep per CU
    res = np.zeros(4); gaps = []; perslot = []
    for c in np.unique(cu):
        m = cu == c
        ev = sorted([(s, 1) for s in t0[m]] + [(e, -1) for e in t1[m]])
        cur, last = 0, 0.0
        for t, d in ev:
            res[min(cur, 3)] += t - last
            last = t; cur += d
        res[0] += span - last
        # greedy slots: a workgroup takes the slot that was freed last before its start
        free = []
        for s, e in sorted(zip(t0[m], t1[m])):
            cand = [f for f in free if f <= s + 1e-9]
            if cand:
                f = max(cand); free.remove(f); gaps.append(s - f)
            free.append(e)
        perslot.append(m.sum())
    res /= span * ncu
    gaps = np.array(gaps)
    kc = a[:, 1] - a[:, 0]
    print(f"column {j:2d}: {len(a):6d} workgroups on {ncu} CUs ({np.min(perslot)}..{np.max(perslot)} per CU)  span {span:8.1f} us  "
          f"lifetime mean {life.mean():6.1f} median {np.median(life):6.1f} p95 {np.percentile(life, 95):6.1f} us  "
          f"sum(lifetime)/(2 x CUs x span) = {life.sum() / (2 * ncu * span):.3f}")
    print(f"           CU residency over the span: 0 wg {res[0]:.3f}  1 wg {res[1]:.3f}  2 wg {res[2]:.3f}  3+ {res[3]:.3f}   "
          f"slot gap between consecutive workgroups: mean {gaps.mean():5.2f} median {np.median(gaps):5.2f} p95 {np.percentile(gaps, 95):5.2f} us "
          f"(n {len(gaps)})   first start spread {np.percentile(t0, 2):.1f} us (p2)  last 2 % of ends after {np.percentile(t1, 98):.1f} us")
    if j:
        print(f"           cycles: K-loop mean {kc.mean():8.0f} median {np.median(kc):8.0f}  total mean {a[:, 3].mean():8.0f} median {np.median(a[:, 3]):8.0f}")
