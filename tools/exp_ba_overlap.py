"""Experiment: how much would overlapping thx_ba_schur (HBM-bound) with the tile-sparse factorisation of the reduced system
(latency / MFMA-bound at batch 256) buy?  Upper bound: the two run on two streams with NO dependency (the Schur kernels write
a second S buffer).  usage: python tools/exp_ba_overlap.py [cams] [points] [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import theseus_amd as th
from theseus_amd.utils.synthetic_ba import make_ba_objective

C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Np = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dt = torch.float32
obj, meta = make_ba_objective(C, Np, B, dtype=dt)
opt = th.LevenbergMarquardt(obj, max_iterations=3, abs_err_tolerance=0.0, rel_err_tolerance=0.0)
with torch.no_grad():
    th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(damping=1e-2, adaptive_damping=True, ellipsoidal_damping=True))
torch.cuda.synchronize()
solver, lin = opt.linear_solver, opt.linear_solver.linearization
p, K = lin.packed, opt.linear_solver.K
lam = torch.full((B,), 1e-2, dtype=dt, device="cuda")
S2, rhs2, Hinv2, tvec2, info2 = (x.clone() for x in (solver.S, solver.rhs, solver.Hinv, solver.tvec, solver.info_pts))
lin._assemble()
K.ba_schur(p.dstruct, lin.Hcc, lin.Hpp, lin.W, lin.gd, lam, True, 1e-8, solver.S, solver.rhs, solver.Hinv, solver.tvec, solver.info_pts)
torch.cuda.synchronize()
side = torch.cuda.Stream()
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731


def schur(S, rhs, Hinv, tvec, info):
    K.ba_schur(p.dstruct, lin.Hcc, lin.Hpp, lin.W, lin.gd, lam, True, 1e-8, S, rhs, Hinv, tvec, info)


def factor():
    K.chol_factor_sparse(solver.S, p.nc, None, False, 1e-8, solver.L, solver.panels, solver.info_chol, solver.pattern, rhs=solver.rhs, y=solver._y)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        schur(S2, rhs2, Hinv2, tvec2, info2)
    factor()
    torch.cuda.current_stream().wait_stream(side)


def both_asm():   # + the linearization's assembly kernels on the side stream too
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        lin._assemble()
        schur(S2, rhs2, Hinv2, tvec2, info2)
    factor()
    torch.cuda.current_stream().wait_stream(side)


t_s, t_f = timed(lambda: schur(S2, rhs2, Hinv2, tvec2, info2)), timed(factor)
t_a = timed(lambda: lin._assemble())
print(f"alone: ba_schur {t_s:.3f} ms, chol_factor_sparse {t_f:.3f} ms, ba_assemble {t_a:.3f} ms; sum schur + factor {t_s + t_f:.3f} ms")
t_b = timed(both)
print(f"two streams, no dependency: schur || factor {t_b:.3f} ms  (saves {t_s + t_f - t_b:.3f} ms of {t_s + t_f:.3f})")
t_c = timed(both_asm)
print(f"two streams: (assemble + schur) || factor {t_c:.3f} ms  (saves {t_a + t_s + t_f - t_c:.3f} ms of {t_a + t_s + t_f:.3f})")
