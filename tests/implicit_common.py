"""Shared by the CPU (stand-in kernels) and GPU (HIP kernels) implicit-backward tests: build the objective of an
implicit golden fixture with requires_grad leaves, run TheseusLayer(backward_mode="implicit"), return grads."""
import numpy as np
import torch

from tests.helpers import golden_problem


def run_implicit(th, g, device, kernels=None):
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    _, _, kw = golden_problem(g)
    kw.pop("gauss_newton")
    P = int(g["P"])
    leaves = dict(meas=t(g["meas"]).requires_grad_(True), w_between=t(g["w_between"]).requires_grad_(True),
                  prior_target=t(g["prior_target"]).requires_grad_(True),
                  w_prior=t(g["w_prior"])[:, :, :1].clone().requires_grad_(True))
    obj = th.Objective(dtype=leaves["meas"].dtype)
    poses0 = t(g["poses0"])
    G = {"SE2": th.SE2, "SO3": th.SO3}.get(str(g["group"]) if "group" in g else "SE3", th.SE3)
    poses = [G(tensor=poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
    for k in range(g["edges"].shape[0]):
        i, j = g["edges"][k].tolist()
        cw = th.DiagonalCostWeight(th.Variable(leaves["w_between"][:, k], name=f"w_{k}"))
        obj.add(th.Between(poses[i], poses[j], G(tensor=leaves["meas"][:, k], name=f"meas_{k}"), cw,
                           name=f"between_{k}"))
    for k in range(g["prior_idx"].shape[0]):
        sw = th.ScaleCostWeight(th.Variable(leaves["w_prior"][:, k], name=f"pw_{k}"))
        obj.add(th.Difference(poses[int(g["prior_idx"][k])],
                              G(tensor=leaves["prior_target"][:, k], name=f"prior_target_{k}"), sw, name=f"prior_{k}"))
    lkw = dict(kernels=kernels) if kernels is not None else None
    opt = th.LevenbergMarquardt(obj, linearization_kwargs=lkw, max_iterations=kw.pop("max_iterations"),
                                step_size=kw.pop("step_size"), abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    layer = th.TheseusLayer(opt)
    sol, info = layer.forward(None, optimizer_kwargs=dict(backward_mode="implicit", track_err_history=True, **kw))
    final = torch.stack([sol[f"pose_{k}"] for k in range(P)], 1)
    loss = (t(g["coef"]) * final).sum()
    loss.backward()
    return final.detach().cpu(), float(loss.detach()), {k: v.grad.detach().cpu() for k, v in leaves.items()}, info, opt, layer


def check_against_reference(g, final, loss, grads, rel=2e-6):
    np.testing.assert_allclose(final.numpy(), g["final"], rtol=0, atol=1e-7)
    assert abs(loss - float(g["loss"])) < 1e-6 * max(1.0, abs(float(g["loss"])))
    for key, ref in (("meas", "grad_meas"), ("w_between", "grad_w_between"), ("prior_target", "grad_prior_target"),
                     ("w_prior", "grad_w_prior")):
        want = g[ref]
        np.testing.assert_allclose(grads[key].numpy(), want, rtol=0, atol=rel * np.abs(want).max(), err_msg=key)
