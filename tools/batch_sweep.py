"""The dense solver on the headline graph (256 poses / 1024 edges, fp32) over the whole batch range, default schedules: LM iteration
time, factorisation (+ fused forward substitution) time and its fraction of the fp32 MFMA peak per batch size.
usage: python tools/batch_sweep.py [batches]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import theseus_amd as th
from theseus_amd.utils import synthetic as syn

P, E, iters, dtype, dev = 256, 1024, 10, torch.float32, "cuda"
n = 6 * P
batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,16,32,64,128,256,512,1024,2048,4096").split(",")]
edges = syn.pose_graph_topology(P, E, topology_seed=0)
print(f"{'batch':>6s} {'ms/LM iteration':>16s} {'problem-iter/s':>15s} {'factor ms':>10s} {'TFLOP/s':>8s} {'of peak':>8s}  schedule")
for B in batches:
    inputs = syn.input_dict(syn.make_pose_graph_tensors(edges, P, B, dtype=dtype, device=dev, seed=77 + B))
    obj = syn.build_pose_graph_objective(edges, P, dtype=dtype, device=dev)
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipCholeskySolver, max_iterations=iters, abs_err_tolerance=0.0,
                                rel_err_tolerance=0.0, step_size=1.0)
    layer = th.TheseusLayer(opt)
    okw = dict(damping=1e-3, track_err_history=True)
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
        s, lin = opt.linear_solver, opt.linear_solver.linearization
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.factorize(1e-3, True, 1e-8, rhs=lin.g)
        ev0.record()
        for _ in range(5):
            s.factorize(1e-3, True, 1e-8, rhs=lin.g)
        ev1.record()
        torch.cuda.synchronize()
        fac = ev0.elapsed_time(ev1) / 5
    tf = B * n ** 3 / 3.0 / (fac * 1e-3) / 1e12
    sched = "right-looking" if B <= int(os.environ.get("THX_CHOL_RL_MAX_BATCH", "64")) else ("left-looking" if B < 128 else ("left-looking, column pairs" if B < 1024 else
                                                                            "left-looking, column pairs, two half-batch streams"))
    print(f"{B:6d} {best:16.3f} {B / best * 1e3:15.0f} {fac:10.3f} {tf:8.1f} {tf / 157.3:8.3f}  {sched}", flush=True)
    del sol, info, layer, opt, obj, inputs
    torch.cuda.empty_cache()
