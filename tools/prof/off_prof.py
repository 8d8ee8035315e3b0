"""In-kernel cycle stamps of chol_offdiag_f32_kernel (library built with -DTHX_OFF_PROF: tools/variants.sh prof32:"-DTHX_OFF_PROF",
run with THESEUS_HIP_LIB=<that .so>): entry -> first chunk issued, K-loop, P = H - sum, substitution, store; wall (100 MHz).
Both instantiations: dense H frames and the LM loop's block-compact H.  usage: python tools/prof/off_prof.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import theseus_amd as th
from theseus_amd.kernels import default_kernels, round_up
from theseus_amd.utils import synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n, dt = 1536, torch.float32
K = default_kernels(); ld = round_up(n, 32)
nt = n // 128
TILES = ((1, 0), (11, 0), (2, 1), (11, 1), (3, 2), (6, 5), (11, 5), (11, 10))


def report(L, tag):
    print(f"== {tag}: cycles per phase, median over the batch")
    for (i, j) in TILES:
        m = L[:, 128 * i, 128 * j + 1:128 * j + 7].double().cpu().median(0).values
        nk = 128 * j // 32
        print(f"tile ({i:2d},{j:2d}): prologue {m[0]:7.0f}  kloop +{m[1] - m[0]:8.0f} ({(m[1] - m[0]) / max(nk, 1):5.0f}/chunk)  H +{m[5] - m[1]:7.0f}  "
              f"trsm +{m[2] - m[5]:7.0f}  store +{m[3] - m[2]:7.0f}  total {m[3]:8.0f} = {m[4] * 10 / 1e3:6.1f} us  (clock {m[3] / (m[4] * 10e-9) / 1e9:.2f} GHz)")


gen = torch.Generator(device="cuda").manual_seed(0)
H = torch.empty(B, ld, ld, dtype=dt, device="cuda"); H.uniform_(-1, 1, generator=gen)
H.diagonal(dim1=1, dim2=2).add_(float(n))
L = torch.zeros_like(H); P = torch.empty(B, nt, 128, 128, dtype=dt, device="cuda")
info = torch.empty(B, dtype=torch.int32, device="cuda"); lam = torch.full((B,), 1e-3, dtype=dt, device="cuda")
for _ in range(2):
    K.chol_factor(H, n, lam, False, 1e-8, L, P, info)
torch.cuda.synchronize()
report(L, "dense H frames (chol_offdiag_f32_kernel<false>)")
del H, L, P
torch.cuda.empty_cache()

edges = syn.pose_graph_topology(256, 1024, topology_seed=0)
obj = syn.build_pose_graph_objective(edges, 256, dtype=dt, device="cuda:0")
opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipCholeskySolver, max_iterations=1, abs_err_tolerance=0.0, rel_err_tolerance=0.0)
obj.update(syn.input_dict(syn.make_pose_graph_tensors(edges, 256, B, dtype=dt, device="cuda:0", seed=1)))
lin, solver = opt.linear_solver.linearization, opt.linear_solver
with torch.no_grad():
    lin.packed.sync(deep=True)
    for _ in range(2):
        lin.linearize()
        solver.factorize(1e-3, False, 1e-8, rhs=lin.g)
torch.cuda.synchronize()
report(solver.L, "block-compact H (chol_offdiag_f32_kernel<true>, the LM loop's)")
