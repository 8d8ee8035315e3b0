"""BASELINE.json configs[0] on theseus_amd's own API: the curve fit of the reference's examples/simple_example.py -- y = v exp(x),
one AutoDiffCostFunction on a 1-d Vector, Gauss-Newton inner loop, implicit backward through TheseusLayer to LEARN the abscissae
x with Adam -- with the inner loop on the HIP path (theseus_amd/euclidean.py: thx_block_assemble + the tiled Cholesky).
usage: python examples/curve_fit_autodiff.py [--batch 16] [--points 20] [--epochs 20] [--device cuda]
       (--kernels module:Class substitutes the kernel set -- the tests' CPU stand-in, for machines without a GPU)
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import theseus_amd as th  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--points", type=int, default=20)
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--kernels", default=None)
    a = ap.parse_args()
    dt, dev = torch.float64, torch.device(a.device)
    kernels = None
    if a.kernels:
        mod, cls = a.kernels.split(":")
        kernels = getattr(importlib.import_module(mod), cls)()
    abscissae = torch.linspace(-1, 1, a.points, dtype=dt, device=dev).repeat(a.batch, 1)
    amplitude = 0.5 * torch.ones(a.batch, 1, dtype=dt, device=dev)
    samples = amplitude * abscissae.exp()

    x = th.Variable(abscissae.clone(), name="x")
    y = th.Variable(samples, name="y")
    v = th.Vector(tensor=torch.ones(a.batch, 1, dtype=dt, device=dev), name="v")

    def residual(optim_vars, aux_vars):
        xs, ys = aux_vars
        return ys.tensor - optim_vars[0].tensor * xs.tensor.exp()

    objective = th.Objective(dtype=dt)
    objective.add(th.AutoDiffCostFunction([v], residual, a.points, aux_vars=[x, y],
                                          cost_weight=th.ScaleCostWeight(torch.ones(1, 1, dtype=dt, device=dev))))
    inner = th.GaussNewton(objective, max_iterations=10, linearization_kwargs=dict(kernels=kernels) if kernels else None)
    layer = th.TheseusLayer(inner)

    phi = torch.nn.Parameter(abscissae + 0.1)          # the outer variable: a shifted guess of the abscissae
    outer = torch.optim.Adam([phi], lr=1e-3)
    for epoch in range(a.epochs):
        outer.zero_grad()
        solution, info = layer.forward({"x": phi.clone(), "v": torch.ones(a.batch, 1, dtype=dt, device=dev)},
                                       optimizer_kwargs={"backward_mode": "implicit"})
        loss = torch.nn.functional.mse_loss(solution["v"], amplitude)
        loss.backward()
        outer.step()
        print(f"epoch {epoch:2d}: outer loss {loss.item():.6e}  (inner iterations {int(info.iters_done)})")


if __name__ == "__main__":
    main()
