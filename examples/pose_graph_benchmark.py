"""examples/pose_graph/pose_graph_benchmark.py of the reference on the HIP path: read a SLAM-3D g2o file, build one
Between per edge + a weak prior on pose 0, run Levenberg-Marquardt (dense HIP linearization + batched dense Cholesky),
print the objective and the time.  usage: python examples/pose_graph_benchmark.py FILE.g2o [--dtype float32|float64]
[--iters 10] [--replicas B]   (--replicas solves B copies of the graph as one batch: the layout the kernels are built for)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import theseus_amd as th  # noqa: E402
from theseus_amd.utils import g2o  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--dtype", default="float64", choices=["float32", "float64"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--replicas", type=int, default=1)
    a = ap.parse_args()
    dtype = getattr(torch, a.dtype)
    _, verts, edges = g2o.read_3D_g2o_file(a.file, dtype=torch.float64)
    if a.replicas > 1:
        for v in verts + [e.relative_pose for e in edges]:
            v.tensor = v.tensor.expand(a.replicas, -1, -1).contiguous()
    g2o.PoseGraphDataset(verts, edges).to(device="cuda", dtype=dtype)
    objective = g2o.pose_graph_objective(verts, edges, dtype=dtype)
    optimizer = th.LevenbergMarquardt(objective, max_iterations=a.iters, step_size=1, linear_solver_cls=th.HipCholeskySolver,
                                      abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        start.record()
        info = optimizer.optimize(track_err_history=True, verbose=False)
        end.record()
    torch.cuda.synchronize()
    print(f"{len(verts)} poses, {len(edges)} edges, batch {a.replicas}: objective {info.err_history[0, 0].item():.6g} -> "
          f"{info.err_history[0, -1].item():.6g} in {start.elapsed_time(end):.1f} ms ({a.iters} LM iterations, {a.dtype})")


if __name__ == "__main__":
    main()
