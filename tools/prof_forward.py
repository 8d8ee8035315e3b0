"""Host profile of TheseusLayer.forward at the headline size (where does the per-optimize() host time go?).
usage: python tools/prof_forward.py [batch] [iters]"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import theseus_amd as th
from theseus_amd.utils import synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
P, E = 256, 1024
edges = syn.pose_graph_topology(P, E, topology_seed=0)
inputs = syn.input_dict(syn.make_pose_graph_tensors(edges, P, B, dtype=torch.float32, device="cuda", seed=1))
obj = syn.build_pose_graph_objective(edges, P, dtype=torch.float32, device="cuda")
opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipCholeskySolver, max_iterations=iters, abs_err_tolerance=0.0,
                            rel_err_tolerance=0.0)
layer = th.TheseusLayer(opt)
kw = dict(damping=1e-3)
with torch.no_grad():
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
print(f"forward({iters} iterations, batch {B}): host returned after {(t1 - t0) * 1e3:.1f} ms, device done after {(t2 - t0) * 1e3:.1f} ms")
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
