"""Bundle adjustment at BASELINE.json configs[3] size on one GPU: LM iterations/s and the per-phase times.
usage: python tools/bench_ba.py [cams] [points] [batch] [dtype] [iters]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import theseus_amd as th
from theseus_amd.utils.synthetic_ba import make_ba_objective

C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Np = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dt = {"f32": torch.float32, "f64": torch.float64}[sys.argv[4] if len(sys.argv) > 4 else "f32"]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
t0 = time.time()
obj, meta = make_ba_objective(C, Np, B, dtype=dt)
print(f"objective built in {time.time() - t0:.1f} s: {meta['num_cams']} cams, {meta['num_points']} points, {meta['num_obs']} obs, n = {meta['n']}")
ordering = os.environ.get("BENCH_BA_ORDERING", "auto")   # camera order of the reduced system (HipSchurSolver(ordering=...))
opt = th.LevenbergMarquardt(obj, max_iterations=iters, abs_err_tolerance=0.0, rel_err_tolerance=0.0,
                            linear_solver_kwargs=dict(ordering=ordering))
layer = th.TheseusLayer(opt)
kw = dict(damping=1e-2, adaptive_damping=True, ellipsoidal_damping=True, track_err_history=True)
t0 = time.time()
with torch.no_grad():
    sol, info = layer.forward(None, optimizer_kwargs=kw)
torch.cuda.synchronize()
print(f"first optimize (incl. packing): {time.time() - t0:.2f} s; mean error {info.err_history[:, 0].mean():.1f} -> {info.err_history[:, -1].mean():.3f}")
solver, lin = opt.linear_solver, opt.linear_solver.linearization
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
phases = {}
lam = torch.full((B,), 1e-2, dtype=dt, device="cuda")


def timed(name, fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    phases[name] = e0.elapsed_time(e1) / reps


p = lin.packed
timed("ba_assemble", lambda: lin._assemble())
if solver.levels:
    t = solver._level_t
    K = solver.K
    timed("ba_schur_blocks", lambda: K.ba_schur_blocks(p.dstruct, lin.Hcc, lin.Hpp, lin.W, lin.gd, lam, True, 1e-8, solver.Sc, t["diag_blk"], t["blk_dst"], solver.rhs, solver.Hinv, solver.tvec, solver.info_pts))
    timed("vec_gather", lambda: K.vec_gather(solver.rhs, solver._xp, t["col_of_pad"]))
    timed("chol_factor(S)", lambda: K.chol_factor_levels(solver._level_layout, solver.Sc, None, False, 1e-8, solver.L, solver.panels, solver.info_chol, solver.pattern, rhs=solver._xp, y=solver._yp))
    timed("chol_backward", lambda: K.chol_solve_levels(solver.L, solver.panels, solver._yp, solver._xp, solver.pattern, which=1))
    timed("ba_backsub", lambda: K.ba_backsub(p.dstruct, lin.W, solver.Hinv, solver.tvec, solver.delta))
    dense_ms = float("nan")
    print(f"level mode: ordering {solver.ordering_info.get('method')}, {solver.pattern.tree_levels} tree levels ({solver.pattern.nlevels} launch levels, two streams: {solver.pattern.two_streams}) over {solver.pattern.ntiles} tiles, "
          f"L tiles {solver.pattern.l_tiles}; candidates {solver.ordering_info.get('candidates')}")
else:
  timed("ba_schur", lambda: solver.K.ba_schur(p.dstruct, lin.Hcc, lin.Hpp, lin.W, lin.gd, lam, True, 1e-8, solver.S, solver.rhs, solver.Hinv, solver.tvec, solver.info_pts))
  timed("chol_factor(S) dense", lambda: solver.K.chol_factor(solver.S, p.nc, None, False, 1e-8, solver.L, solver.panels, solver.info_chol, rhs=solver.rhs, y=solver._y))
  dense_ms = phases.pop("chol_factor(S) dense")   # for comparison only: the solver factorises along the tile pattern of S
  if solver.sparse:
      timed("chol_factor(S)", lambda: solver.K.chol_factor_sparse(solver.S, p.nc, None, False, 1e-8, solver.L, solver.panels, solver.info_chol, solver.pattern, rhs=solver.rhs, y=solver._y))
  else:
      phases["chol_factor(S)"] = dense_ms
  if solver.sparse:
      timed("chol_backward", lambda: solver.K.chol_solve_sparse(solver.L, p.nc, solver.panels, solver._y, solver._dc, solver.pattern, backward_only=True))
  else:
      timed("chol_backward", lambda: solver.K.chol_solve_backward(solver.L, p.nc, solver.panels, solver._y, solver._dc))
  timed("ba_backsub", lambda: solver.K.ba_backsub(p.dstruct, lin.W, solver.Hinv, solver.tvec, solver.delta))
timed("ba_error", lambda: p.error_metric())
spare = p.alloc_state()
timed("retract", lambda: p.retract(solver.delta, 1.0, None, spare))
if os.environ.get("BENCH_BA_NOGC"):
    import gc
    gc.disable()
e0, e1 = ev(), ev()
with torch.no_grad():
    obj.update()
    torch.cuda.synchronize()
    fv0, w0 = solver.factor_version, time.perf_counter()
    e0.record()
    if os.environ.get("BENCH_BA_CPROFILE"):  # where the host time of optimize() goes
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        info = opt.optimize(**kw)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    else:
        info = opt.optimize(**kw)
    e1.record()
    host_ms = (time.perf_counter() - w0) * 1e3   # host time until optimize() returns (launches may still be queued)
torch.cuda.synchronize()
wall_ms = (time.perf_counter() - w0) * 1e3
solves = solver.factor_version - fv0
ms = e0.elapsed_time(e1) / max(info.iters_done, 1)
print(f"optimize(): {info.iters_done} accepted iterations, {solves} linear solves; device span {e0.elapsed_time(e1):.1f} ms, "
      f"host returned after {host_ms:.1f} ms, wall {wall_ms:.1f} ms -> {e0.elapsed_time(e1) / max(solves, 1):.2f} ms per solve")
nc = p.nc
fl = B * nc ** 3 / 3
print("phases (ms):", {k: round(v, 3) for k, v in phases.items()})
per_solve = sum(phases.values())
print(f"one linearize + solve + retract + error: {per_solve:.2f} ms of kernels -> {B / per_solve * 1e3:.0f} problem-iterations/s; "
      f"LM loop {ms:.2f} ms per ACCEPTED iteration (all-rejected retries re-solve, nonlinear_least_squares.py:358-365); "
      f"Schur system {nc}^2: L has {solver.pattern.l_tiles} of {solver.pattern.ntiles * (solver.pattern.ntiles + 1) // 2} tiles, "
      f"{solver.pattern.flops / solver.pattern.dense_flops:.3f} of the dense flops; tile-sparse Cholesky "
      f"{B * solver.pattern.flops / phases['chol_factor(S)'] / 1e9:.1f} TFLOP/s executed; the dense factorisation of the same S: "
      f"{dense_ms:.2f} ms = {fl / dense_ms / 1e9:.1f} TFLOP/s")
