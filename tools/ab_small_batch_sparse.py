"""256 poses / 1024 edges at batch 32 ... 256: the dense solver against HipSparseCholeskySolver(ordering="rcm") -- the column
schedule with its look-ahead (head tile on its own launch, the next diagonal phase beside the rest of the column) on an almost
dense pattern: does the look-ahead pay on dense systems at these batch sizes?  usage: python tools/ab_small_batch_sparse.py [batches]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import theseus_amd as th
from theseus_amd.utils import synthetic as syn

P, E, iters, dtype, dev = 256, 1024, 10, torch.float32, "cuda"
n = 6 * P
batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32,64,128,256").split(",")]
edges = syn.pose_graph_topology(P, E, topology_seed=0)
for B in batches:
    inputs = syn.input_dict(syn.make_pose_graph_tensors(edges, P, B, dtype=dtype, device=dev, seed=77 + B))
    for name, cls, kw in (("dense", th.HipCholeskySolver, {}), ("sparse rcm (look-ahead)", th.HipSparseCholeskySolver, dict(ordering="rcm")),
                          ("sparse auto", th.HipSparseCholeskySolver, dict(ordering="auto", batch_hint=B))):
        obj = syn.build_pose_graph_objective(edges, P, dtype=dtype, device=dev)
        opt = th.LevenbergMarquardt(obj, linear_solver_cls=cls, linear_solver_kwargs=kw, max_iterations=iters, abs_err_tolerance=0.0,
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
        s = opt.linear_solver
        extra = f"; ordering {getattr(s, 'ordering_info', {}).get('method')}, levels {getattr(s, 'levels', None)}" if hasattr(s, "ordering_info") else ""
        print(f"batch {B:4d} {name:26s}: {best:7.3f} ms / LM iteration; error {float(info.err_history[:, 0].mean()):.1f} -> "
              f"{float(info.err_history[:, info.iters_done].mean()):.4f}{extra}", flush=True)
        del sol, info, layer, opt, obj
