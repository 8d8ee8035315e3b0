"""Dense Cholesky at the reference's batch sizes (256 poses / 1024 edges, batch 8 ... 256): the left-looking schedule against the
right-looking one (thx_chol_schedule.right_looking_max_batch) and column pairs on / off, same process, same inputs.
usage: python tools/ab_small_batch.py [batches, default 8,16,32,64,128,256] [f32|f64]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import theseus_amd as th
from theseus_amd.utils import synthetic as syn

P, E, iters, dev = 256, 1024, 10, "cuda"
dtype = {"f32": torch.float32, "f64": torch.float64}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
PEAK = 157.3 if dtype == torch.float32 else 78.6
n = 6 * P
batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,16,32,64,128,256").split(",")]
edges = syn.pose_graph_topology(P, E, topology_seed=0)
for B in batches:
    inputs = syn.input_dict(syn.make_pose_graph_tensors(edges, P, B, dtype=dtype, device=dev, seed=77 + B))
    for name, rl, pairs in (("left-looking, pairs", 0, 1), ("left-looking, no pairs", 0, 0), ("right-looking", 1 << 20, 1)):
        obj = syn.build_pose_graph_objective(edges, P, dtype=dtype, device=dev)
        opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipCholeskySolver, max_iterations=iters, abs_err_tolerance=0.0,
                                    rel_err_tolerance=0.0, step_size=1.0)
        K = opt.linear_solver.K
        prev = (K.chol_right_looking_max_batch(rl), K.chol_column_pairs(pairs))
        layer = th.TheseusLayer(opt)
        okw = dict(damping=1e-3, track_err_history=True)
        try:
            with torch.no_grad():
                opt.set_params(max_iterations=2)
                layer.forward(inputs, optimizer_kwargs=okw)
                opt.set_params(max_iterations=iters)
                best = None
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    sol, info = layer.forward(inputs, optimizer_kwargs=okw)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / info.iters_done * 1e3
                    best = dt if best is None else min(best, dt)
                # the factorisation alone
                s, lin = opt.linear_solver, opt.linear_solver.linearization
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.factorize(1e-3, True, 1e-8, rhs=lin.g)
                ev0.record()
                for _ in range(10):
                    s.factorize(1e-3, True, 1e-8, rhs=lin.g)
                ev1.record()
                torch.cuda.synchronize()
                fac = ev0.elapsed_time(ev1) / 10
        finally:
            K.chol_right_looking_max_batch(prev[0])
            K.chol_column_pairs(prev[1])
        tf = B * n ** 3 / 3.0 / (fac * 1e-3) / 1e12
        print(f"batch {B:4d} {name:24s}: {best:7.3f} ms / LM iteration, factor + forward {fac:7.3f} ms = {tf:6.1f} TFLOP/s "
              f"({tf / PEAK:.3f} of peak); error {float(info.err_history[:, 0].mean()):.1f} -> "
              f"{float(info.err_history[:, info.iters_done].mean()):.4f}", flush=True)
        del sol, info, layer, opt, obj
