"""The fp64 half-tile off-diagonal kernel (THX_F64_HALF_MAX_KTILES, read once per process) against the product: the factor of the
reference-size fixture in two subprocesses, block-compact and dense H, compared bit for bit.  usage: python tools/cmp_f64_half.py"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from tests.test_gpu_block_hessian import _assembled
from theseus_amd.kernels import default_kernels
K = default_kernels()
s, hb, dhb, H, gv, Hc, g2, n, ld = _assembled(K, "pg_full_f64_lm")
B, nt = H.shape[0], (n + 127) // 128
lam = torch.full((B,), 1e-3, dtype=H.dtype, device="cuda")
out = {}
for compact in (True, False):
    L = torch.zeros_like(H); panels = torch.zeros(B, nt, 128, 128, dtype=H.dtype, device="cuda")
    info = torch.empty(B, dtype=torch.int32, device="cuda"); y = torch.empty_like(gv)
    if compact: K.chol_factor_hblocks(dhb, Hc, n, lam, True, 1e-8, L, panels, info, rhs=gv, y=y)
    else: K.chol_factor(H, n, lam, True, 1e-8, L, panels, info, rhs=gv, y=y)
    out[compact] = (torch.tril(L[:, :n, :n]).cpu(), y.cpu(), info.cpu())
torch.save(out, sys.argv[1])
''' % ROOT
res = {}
for half in ("0", "12"):
    env = dict(os.environ, THX_F64_HALF_MAX_KTILES=half)
    path = f"/tmp/f64_half_{half}.pt"
    subprocess.run([sys.executable, "-c", CHILD, path], check=True, env=env, cwd=ROOT)
    import torch
    res[half] = torch.load(path)
for compact in (True, False):
    (La, ya, ia), (Lb, yb, ib) = res["0"][compact], res["12"][compact]
    print("block-compact H" if compact else "dense H       ", "info", int(ia.abs().sum()), int(ib.abs().sum()), "L equal", torch.equal(La, Lb), "y equal",
          torch.equal(ya, yb), "max |dL|", float((La - Lb).abs().max()))
