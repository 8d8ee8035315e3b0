"""Host-side cost of one pose-graph TheseusLayer.forward(inputs) (256 poses / 1024 edges) with every kernel a no-op."""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import theseus_amd as th


class Null:
    name = "null"

    def __getattr__(self, k):
        def f(*a, **kw):
            return None
        return f


K = Null()
P, E, B = 256, 1024, 8
dt = torch.float32
gen = torch.Generator().manual_seed(0)
eye = torch.eye(3, 4, dtype=dt).unsqueeze(0).repeat(B, 1, 1)
poses = [th.SE3(tensor=eye.clone(), name=f"VERTEX_SE3__{k}") for k in range(P)]
obj = th.Objective(dtype=dt)
w = th.DiagonalCostWeight(th.Variable(torch.ones(1, 6, dtype=dt), name="w"))
inputs = {}
for e in range(E):
    i, j = (e % P, (e * 7 + 1) % P)
    if i == j:
        j = (j + 1) % P
    m = th.SE3(tensor=eye.clone(), name=f"EDGE_SE3__{e}")
    inputs[m.name] = eye.clone()
    obj.add(th.Between(poses[i], poses[j], m, w, name=f"between_{e}"))
obj.add(th.Difference(poses[0], th.SE3(tensor=eye.clone(), name="prior_t"), th.ScaleCostWeight(torch.tensor(1e-3)), name="prior"))
for k in range(P):
    inputs[f"VERTEX_SE3__{k}"] = eye.clone()
opt = th.LevenbergMarquardt(obj, max_iterations=5, abs_err_tolerance=0.0, rel_err_tolerance=0.0, linearization_kwargs=dict(kernels=K))
layer = th.TheseusLayer(opt)
kw = dict(damping=1e-3, track_err_history=True)
with torch.no_grad():
    layer.forward(inputs, optimizer_kwargs=kw)
    for _ in range(3):
        t0 = time.perf_counter()
        layer.forward(inputs, optimizer_kwargs=kw)
        print("forward host ms", (time.perf_counter() - t0) * 1e3)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        layer.forward(inputs, optimizer_kwargs=kw)
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
