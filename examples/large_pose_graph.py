"""A LARGE chain-like SE3 pose graph (odometry + local loop closures, shuffled labels) at the reference's sweep sizes
(evaluations/pose_graph_synthetic.sh: up to 4096 poses, batch 8 - 256) with the tile-sparse solver: tile-level nested dissection
of the variables, one diagonal + one off-diagonal launch per level of the tile elimination tree (theseus_amd/sparse.py,
thx_chol_factor_levels) -- the role BaspachoSparseSolver plays in the reference.

    python examples/large_pose_graph.py [--poses 4096] [--batch 64] [--iters 10] [--ordering auto|nd|rcm]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import theseus_amd as th
from theseus_amd.utils.synthetic import chain_graph_topology


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--ordering", default="auto")
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    a = ap.parse_args()
    dtype, dev, P, B = getattr(torch, a.dtype), "cuda", a.poses, a.batch
    edges = chain_graph_topology(P, stride=7, span=5, seed=2, shuffle=True)
    K = th.default_kernels()
    gen = torch.Generator(device=dev).manual_seed(7)
    rnd = lambda n, s: K.se3_exp(s * (2 * torch.rand(n, 6, dtype=dtype, device=dev, generator=gen) - 1))  # noqa: E731
    gt = rnd(B * P, 1.5).view(B, P, 3, 4)                                          # ground truth, one per problem
    poses0 = K.se3_compose(gt.reshape(-1, 3, 4), rnd(B * P, 0.05)).view(B, P, 3, 4)     # noisy initialisation
    obj = th.Objective(dtype=dtype)
    pv = [th.SE3(tensor=poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
    w = th.ScaleCostWeight(torch.tensor(5.0, dtype=dtype, device=dev))
    for k, (i, j) in enumerate(edges):
        z = K.se3_compose(K.se3_compose(K.se3_inverse(gt[:, i].contiguous()), gt[:, j].contiguous()), rnd(B, 0.01))   # noisy measurement
        obj.add(th.Between(pv[i], pv[j], th.SE3(tensor=z, name=f"z_{k}"), w, name=f"between_{k}"))
    obj.add(th.Difference(pv[edges[0][0]], th.SE3(tensor=gt[:, edges[0][0]].clone(), name="anchor"), w, name="prior"))
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipSparseCholeskySolver, linear_solver_kwargs=dict(ordering=a.ordering),
                                max_iterations=a.iters, abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    layer = th.TheseusLayer(opt)
    start = {f"pose_{k}": poses0[:, k].clone() for k in range(P)}
    kw = dict(damping=1e-2, track_err_history=True)
    with torch.no_grad():
        layer.forward(start, optimizer_kwargs=kw)          # (first call: symbolic analysis, packing, buffers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sol, info = layer.forward(start, optimizer_kwargs=kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    s = opt.linear_solver
    pat = s.pattern
    print(f"{P} poses / {len(edges)} edges, n = {6 * P}, batch {B}, {a.dtype}: ordering {s.ordering_info.get('method')}, "
          f"{pat.ntiles} tiles, {pat.l_tiles} tiles of L" + (f", {pat.tree_levels} levels of the tile elimination tree" if s.levels else
                                                           ", column-by-column schedule"))
    print(f"objective {float(info.err_history[:, 0].mean()):.2f} -> {float(info.err_history[:, -1].mean()):.4f} in {dt * 1e3:.1f} ms "
          f"({info.iters_done} LM iterations, {dt / info.iters_done * 1e3:.2f} ms each = {B * info.iters_done / dt:.0f} problem-iterations/s)")


if __name__ == "__main__":
    main()
