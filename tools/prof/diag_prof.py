"""reads the cycle stamps the -DTHX_DIAG_PROF build leaves in the panel tiles"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from theseus_amd.kernels import default_kernels, round_up
n, B, dt = 1536, 4096, torch.float32
K = default_kernels(); ld = round_up(n, 32)
gen = torch.Generator(device="cuda").manual_seed(0)
H = torch.empty(B, ld, ld, dtype=dt, device="cuda"); H.uniform_(-1, 1, generator=gen)
H.diagonal(dim1=1, dim2=2).add_(float(n))
nt = n // 128
L = torch.zeros_like(H); P = torch.empty(B, nt, 128, 128, dtype=dt, device="cuda")
info = torch.empty(B, dtype=torch.int32, device="cuda"); lam = torch.full((B,), 1e-3, dtype=dt, device="cuda")
for _ in range(2):
    K.chol_factor(H, n, lam, False, 1e-8, L, P, info)
torch.cuda.synchronize()
flat = ["start", "kloop", "S done"] + sum([[f"sb{s} pre", f"sb{s} potrf", f"sb{s} st", f"sb{s} inv"] for s in range(4)], []) + ["chol done", "end"]
blocked = ["start", "kloop", "S done"] + sum([[f"sb{s} enter", f"sb{s} potrf+inv"] for s in range(4)], []) + ["chol done", "end"]
for j in (0, 5, 11):
    st = P[:, j, 0, :24].double().cpu()   # (B, 24)
    nst = int(st[0, 0].item())
    names = flat if nst == len(flat) else blocked   # -DTHX_POTRF_FLAT stamps four phases per sub-block, the default two
    d = st[:, 1:nst]
    med = d.median(0).values
    print(f"j={j} nst={nst}: median cycle stamps (delta from previous)")
    prev = 0.0
    for k in range(nst - 1):
        print(f"   {names[k+1]:10s} {med[k].item():10.0f}  (+{med[k].item()-prev:8.0f})")
        prev = med[k].item()
