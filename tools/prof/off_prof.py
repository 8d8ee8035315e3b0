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
for (i, j) in ((1, 0), (11, 0), (6, 5), (11, 5), (11, 10)):
    st = L[:, 128 * i, 128 * j + 1:128 * j + 6].double().cpu()
    med = st.median(0).values
    nk = 128 * j // 32
    print(f"tile ({i},{j}) K-steps {nk}: prefetch {med[0]:.0f}  kloop +{med[1]-med[0]:.0f} ({(med[1]-med[0])/max(nk,1):.0f}/step)  trsm +{med[2]-med[1]:.0f}  store +{med[3]-med[2]:.0f}  total {med[3]:.0f}  wall {med[4]*10:.0f} ns -> shader clock {med[3]/(med[4]*10e-9)/1e9:.3f} GHz")
