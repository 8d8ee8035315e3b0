"""fp64 twin of off_prof.py: stamps = [kloop, H subtract, panel A staged, substitution, stored], wall (100 MHz)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from theseus_amd.kernels import default_kernels, round_up
n, B, dt = 1536, 2048, torch.float64
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
for (i, j) in ((1, 0), (6, 5), (11, 10)):
    st = L[:, 128 * i, 128 * j + 1:128 * j + 7].cpu()
    m = st.median(0).values
    nk = 128 * j // 16
    print(f"tile ({i},{j}) K-steps {nk}: kloop {m[0]:.0f} ({m[0]/max(nk,1):.0f}/step)  H +{m[1]-m[0]:.0f}  panelA +{m[2]-m[1]:.0f}  "
          f"trsm(+panelB) +{m[3]-m[2]:.0f}  store +{m[4]-m[3]:.0f}  total {m[4]:.0f}  clock {m[4]/(m[5]*10e-9)/1e9:.2f} GHz")
