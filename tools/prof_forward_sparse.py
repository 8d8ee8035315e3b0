"""Host profile of ONE TheseusLayer.forward(inputs) of the level-scheduled sparse solver at 4096 poses (the once-per-call cost:
packing the inputs, the read-back, the solution dict).  usage: python tools/prof_forward_sparse.py [poses] [batch] [iters]"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import theseus_amd as th
from tests.test_sparse_solver import chain_graph

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dtype = torch.float32
edges = chain_graph(P, stride=7, span=5, seed=2)
K = th.default_kernels()
gen = torch.Generator(device="cuda").manual_seed(7)
rnd = lambda nn, s: K.se3_exp(s * (2 * torch.rand(nn, 6, dtype=dtype, device="cuda", generator=gen) - 1))  # noqa: E731
gt = rnd(B * P, 1.5).view(B, P, 3, 4)
poses0 = K.se3_compose(gt.reshape(-1, 3, 4), rnd(B * P, 0.05)).view(B, P, 3, 4)
obj = th.Objective(dtype=dtype)
pv = [th.SE3(tensor=poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
w = th.ScaleCostWeight(torch.tensor(5.0, dtype=dtype, device="cuda"))
ident = K.se3_exp(torch.zeros(B, 6, dtype=dtype, device="cuda"))
for k, (i, j) in enumerate(edges):
    obj.add(th.Between(pv[i], pv[j], th.SE3(tensor=ident.clone(), name=f"m_{k}"), w, name=f"b_{k}"))
obj.add(th.Difference(pv[edges[0][0]], th.SE3(tensor=gt[:, edges[0][0]].clone(), name="anchor"), w, name="prior"))
opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipSparseCholeskySolver, max_iterations=iters, abs_err_tolerance=0.0,
                            rel_err_tolerance=0.0)
layer = th.TheseusLayer(opt)
start = {f"pose_{k}": poses0[:, k].clone() for k in range(P)}
kw = dict(damping=1e-2, track_err_history=True)
with torch.no_grad():
    for _ in range(2):
        layer.forward(start, optimizer_kwargs=kw)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    layer.forward(start, optimizer_kwargs=kw)
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"forward({iters} iterations, {P} poses, batch {B}): host returned after {(t1 - t0) * 1e3:.1f} ms, device done after {(t2 - t0) * 1e3:.1f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
