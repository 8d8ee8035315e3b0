"""Micro-benchmark of thx_chol_factor / thx_chol_solve on random SPD batches (HIP events on torch's
current stream).  usage: python tools/bench_chol.py [n] [B] [dtype] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theseus_amd.kernels import default_kernels, round_up

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dt = {"f32": torch.float32, "f64": torch.float64}[sys.argv[3] if len(sys.argv) > 3 else "f32"]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
K = default_kernels()
ld = round_up(n, 32)
gen = torch.Generator(device="cuda").manual_seed(0)
H = torch.empty(B, ld, ld, dtype=dt, device="cuda")
H.uniform_(-1, 1, generator=gen)
H.diagonal(dim1=1, dim2=2).add_(float(n))  # strictly diagonally dominant -> SPD (lower triangle used)
rhs = torch.randn(B, n, dtype=dt, device="cuda", generator=gen)
nt = (n + 127) // 128
L = torch.zeros_like(H)
diagT = torch.empty(B, nt, 128, 128, dtype=dt, device="cuda")
info = torch.empty(B, dtype=torch.int32, device="cuda")
x = torch.empty_like(rhs)
lam = torch.full((B,), 1e-3, dtype=dt, device="cuda")


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)


fmin, favg = timed(lambda: K.chol_factor(H, n, lam, False, 1e-8, L, diagT, info))
smin, savg = timed(lambda: K.chol_solve(L, n, diagT, rhs, x))
y = torch.empty_like(rhs)
x2 = torch.empty_like(rhs)
ffmin, ffavg = timed(lambda: K.chol_factor(H, n, lam, False, 1e-8, L, diagT, info, rhs=rhs, y=y))
bmin, bavg = timed(lambda: K.chol_solve_backward(L, n, diagT, y, x2))
print(f"fused: factor+forward {ffavg:.2f} ms (min {ffmin:.2f}); backward {bavg:.2f} ms (min {bmin:.2f}) = "
      f"{B * n * (n + 1) / 2 * dt.itemsize / bavg / 1e6:.0f} GB/s of tril(L); x2 vs x max diff {(x2 - x).abs().max().item():.3e}")
fl = B * n ** 3 / 3
print(f"n={n} B={B} {dt}: factor {favg:.2f} ms (min {fmin:.2f}) = {fl / favg / 1e9:.1f} TFLOP/s ; "
      f"solve {savg:.2f} ms (min {smin:.2f}) = {B * n * (n + 1) * dt.itemsize / savg / 1e6:.0f} GB/s of L (2 passes over tril)")
assert int(info.abs().sum()) == 0
# residual check on a few problems
Lc = torch.tril(L[:4, :n, :n]).double()
Hc = torch.tril(H[:4, :n, :n]).double()
Hs = Hc + torch.tril(Hc, -1).transpose(1, 2) + 1e-3 * torch.eye(n, device="cuda", dtype=torch.float64)
print("factor resid", ((Lc @ Lc.transpose(1, 2) - Hs).abs().max() / Hs.abs().max()).item(),
      "solve resid", ((Hs @ x[:4].double().unsqueeze(2)).squeeze(2) - rhs[:4].double()).abs().max().item())
