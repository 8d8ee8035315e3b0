"""Host profile of the LM loop at a small batch (256 poses / 1024 edges, batch 8, 40 iterations): where does the host's time per
queued iteration go?  usage: python tools/prof_loop_host.py [batch] [iters]"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import theseus_amd as th
from theseus_amd.utils import synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
P, E = 256, 1024
edges = syn.pose_graph_topology(P, E, topology_seed=0)
inputs = syn.input_dict(syn.make_pose_graph_tensors(edges, P, B, dtype=torch.float32, device="cuda", seed=1))
obj = syn.build_pose_graph_objective(edges, P, dtype=torch.float32, device="cuda")
opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipCholeskySolver, max_iterations=iters, abs_err_tolerance=0.0,
                            rel_err_tolerance=0.0, step_size=1.0)
layer = th.TheseusLayer(opt)
kw = dict(damping=1e-3, track_err_history=True)
with torch.no_grad():
    for _ in range(2):
        layer.forward(inputs, optimizer_kwargs=kw)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    layer.forward(inputs, optimizer_kwargs=kw)
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"forward({iters} iterations, batch {B}): host returned after {(t1 - t0) * 1e3:.1f} ms, device done after {(t2 - t0) * 1e3:.1f} ms "
      f"-> {(t2 - t0) / iters * 1e3:.3f} ms per iteration")
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
