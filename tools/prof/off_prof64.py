"""In-kernel cycle stamps of chol_offdiag_f64_kernel (library built with -DTHX_OFF_PROF64: tools/variants.sh prof64:"-DTHX_OFF_PROF64",
run with THESEUS_HIP_LIB=<that .so>): stamps = [K-loop done, P = H - sum, panel staged, substitution done, stored], wall (100 MHz).
Both instantiations: dense H frames (thx_chol_factor) and the LM loop's block-compact H (thx_chol_factor_hblocks on a pose graph).
usage: python tools/prof/off_prof64.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import theseus_amd as th
from theseus_amd.kernels import default_kernels, round_up
from theseus_amd.utils import synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n, dt = 1536, torch.float64
K = default_kernels(); ld = round_up(n, 32)
nt = n // 128
TILES = ((1, 0), (11, 0), (2, 1), (11, 1), (3, 2), (6, 5), (11, 5), (11, 10))


def report(L, tag):
    print(f"== {tag}: cycles per phase, median over the batch (tile (i, j): K-loop of j tiles of 8 k-chunks)")
    for (i, j) in TILES:
        st = L[:, 128 * i, 128 * j + 1:128 * j + 8].cpu()
        m = st.median(0).values
        sub = L[:, 128 * i, 128 * j + 8:128 * j + 12].cpu().median(0).values   # block-compact H: first barrier, rounds 1..3 begin
        nk = 128 * j // 16
        print(f"tile ({i:2d},{j:2d}): kloop {m[0]:8.0f} ({m[0] / max(nk, 1):5.0f}/chunk)  H +{m[1] - m[0]:7.0f}  panel +{m[2] - m[1]:7.0f}  "
              f"trsm +{m[3] - m[2]:7.0f}  store +{m[4] - m[3]:7.0f}  total {m[4]:8.0f} = {m[5] * 10 / 1e3:6.1f} us  (clock {m[4] / (m[5] * 10e-9) / 1e9:.2f} GHz)"
              + (f"   [H: barrier +{sub[0] - m[0]:.0f}, rounds +{sub[1] - sub[0]:.0f} +{sub[2] - sub[1]:.0f} +{sub[3] - sub[2]:.0f} +{m[1] - sub[3]:.0f}]" if float(sub[0]) > 0 else ""))


gen = torch.Generator(device="cuda").manual_seed(0)
H = torch.empty(B, ld, ld, dtype=dt, device="cuda"); H.uniform_(-1, 1, generator=gen)
H.diagonal(dim1=1, dim2=2).add_(float(n))
L = torch.zeros_like(H); P = torch.empty(B, nt, 128, 128, dtype=dt, device="cuda")
info = torch.empty(B, dtype=torch.int32, device="cuda"); lam = torch.full((B,), 1e-3, dtype=dt, device="cuda")
for _ in range(2):
    K.chol_factor(H, n, lam, False, 1e-8, L, P, info)
torch.cuda.synchronize()
report(L, "dense H frames (chol_offdiag_f64_kernel<false>)")
del H, L, P
torch.cuda.empty_cache()

edges = syn.pose_graph_topology(256, 1024, topology_seed=0)
obj = syn.build_pose_graph_objective(edges, 256, dtype=dt, device="cuda:0")
opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipCholeskySolver, max_iterations=1, abs_err_tolerance=0.0, rel_err_tolerance=0.0)
inputs = syn.input_dict(syn.make_pose_graph_tensors(edges, 256, B, dtype=dt, device="cuda:0", seed=1))
obj.update(inputs)
lin, solver = opt.linear_solver.linearization, opt.linear_solver
with torch.no_grad():
    lin.packed.sync(deep=True)
    for _ in range(2):
        lin.linearize()
        solver.factorize(1e-3, False, 1e-8, rhs=lin.g)
torch.cuda.synchronize()
report(solver.L, "block-compact H (chol_offdiag_f64_kernel<true>, the LM loop's)")
