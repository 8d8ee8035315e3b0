"""Shared by the CPU (oracle / stand-in kernels) and GPU (HIP kernels) tests of MIXED robust objectives: plain, Welsch, Huber and
``flatten_dims=True`` costs inside one role (tests/golden/pg*_f64_mixed_robust.npz, written by oracle/gen_golden.py:
gen_pg_mixed_robust from the REAL reference -- theseus/core/robust_cost_function.py:52-135)."""
import dataclasses

import numpy as np
import torch

from tests.helpers import golden_problem

GRAD_KEYS = (("meas", "grad_meas"), ("w_between", "grad_w_between"), ("prior_target", "grad_prior_target"),
             ("w_prior", "grad_w_prior"), ("log_radius_between", "grad_log_radius_between"),
             ("log_radius_prior", "grad_log_radius_prior"))


def specs(g, role):
    return [str(x) or None for x in g["loss_" + role].tolist()]


def shared_radius(g, role):
    """which costs' log_loss_radius is ONE value for the batch (stored (1, 1) in the reference objective)"""
    lr = g["log_radius_" + role]
    return [bool((lr[:, k] == lr[:1, k]).all()) for k in range(lr.shape[1])]


def effective_radius(log_radius, mu, spec_list):
    """the radius entries as the oracle / the kernels take them: log_loss_radius, + log(gnc_control_val) for the Geman-McClure
    ("gm") costs of a GNCRobustCostFunction fixture (robust_loss.py:96-113 depends on mu * radius only)"""
    if mu is None:
        return log_radius
    gm = torch.tensor([(s or "").split("+")[0] == "gm" for s in spec_list]).view(1, -1, 1)
    return torch.where(gm, log_radius + mu.log(), log_radius)


def mixed_problem(g):
    """fixture -> (oracle PGProblem with per-cost loss specs, poses0, optimizer kwargs)"""
    p, poses0, kw = golden_problem(g)
    mu_b = torch.from_numpy(g["gnc_between"]) if "gnc_between" in g else None
    mu_p = torch.from_numpy(g["gnc_prior"]) if "gnc_prior" in g else None
    p = dataclasses.replace(p, robust_between=specs(g, "between"),
                            log_radius_between=effective_radius(torch.from_numpy(g["log_radius_between"]), mu_b, specs(g, "between")),
                            robust_prior=specs(g, "prior"),
                            log_radius_prior=effective_radius(torch.from_numpy(g["log_radius_prior"]), mu_p, specs(g, "prior")))
    return p, poses0, kw


def run_mixed_implicit(th, g, device, kernels=None, dtype=torch.float64, backward=True, optimizer_kwargs=None):
    """The fixture's objective through ``th`` (theseus_amd's mirror API): LM + TheseusLayer(backward_mode="implicit")."""
    t = lambda a: torch.from_numpy(a).to(device=device, dtype=dtype)  # noqa: E731
    _, _, kw = golden_problem(g)
    kw.pop("gauss_newton")
    P, E, Kp = int(g["P"]), g["edges"].shape[0], g["prior_idx"].shape[0]
    leaves = dict(meas=t(g["meas"]).requires_grad_(backward), w_between=t(g["w_between"]).requires_grad_(backward),
                  prior_target=t(g["prior_target"]).requires_grad_(backward),
                  w_prior=t(g["w_prior"])[:, :, :1].clone().requires_grad_(backward),
                  log_radius_between=t(g["log_radius_between"]).requires_grad_(backward),
                  log_radius_prior=t(g["log_radius_prior"]).requires_grad_(backward))
    if "gnc_between" in g:      # GNCRobustCostFunction fixtures: one control value per cost, shared by the batch
        leaves["gnc_between"] = t(g["gnc_between"]).requires_grad_(backward)
        leaves["gnc_prior"] = t(g["gnc_prior"]).requires_grad_(backward)
    G = {"SE2": th.SE2, "SO3": th.SO3}.get(str(g["group"]), th.SE3)
    LOSS = {"welsch": th.WelschLoss, "huber": th.HuberLoss, "hinge": th.HingeLoss}

    def wrap(cf, spec, radius, shared, nm, mu=None):
        if spec is None:
            return cf
        rv = th.Variable(radius[:1] if shared else radius, name="log_radius_" + nm)
        if spec.split("+")[0] == "gm":
            return th.GNCRobustCostFunction(cf, th.GemanMcClureLoss, rv, th.Variable(mu, name="gnc_" + nm),
                                            flatten_dims=spec.endswith("+flatten"), name="robust_" + nm)
        return th.RobustCostFunction(cf, LOSS[spec.split("+")[0]], rv, flatten_dims=spec.endswith("+flatten"), name="robust_" + nm)

    obj = th.Objective(dtype=dtype)
    poses0 = t(g["poses0"])
    poses = [G(tensor=poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
    sb, sp = shared_radius(g, "between"), shared_radius(g, "prior")
    for k in range(E):
        i, j = g["edges"][k].tolist()
        cw = th.DiagonalCostWeight(th.Variable(leaves["w_between"][:, k], name=f"w_{k}"))
        cf = th.Between(poses[i], poses[j], G(tensor=leaves["meas"][:, k], name=f"meas_{k}"), cw, name=f"between_{k}")
        obj.add(wrap(cf, specs(g, "between")[k], leaves["log_radius_between"][:, k], sb[k], f"between_{k}",
                     leaves["gnc_between"][:, k] if "gnc_between" in leaves else None))
    for k in range(Kp):
        sw = th.ScaleCostWeight(th.Variable(leaves["w_prior"][:, k], name=f"pw_{k}"))
        cf = th.Difference(poses[int(g["prior_idx"][k])], G(tensor=leaves["prior_target"][:, k], name=f"prior_target_{k}"), sw,
                           name=f"prior_{k}")
        obj.add(wrap(cf, specs(g, "prior")[k], leaves["log_radius_prior"][:, k], sp[k], f"prior_{k}",
                     leaves["gnc_prior"][:, k] if "gnc_prior" in leaves else None))
    lkw = dict(kernels=kernels) if kernels is not None else None
    opt = th.LevenbergMarquardt(obj, linearization_kwargs=lkw, max_iterations=kw.pop("max_iterations"), step_size=kw.pop("step_size"),
                                abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    out = dict(obj=obj, opt=opt, leaves=leaves)
    with torch.no_grad():
        obj.update()
        out["err0"] = obj.error_metric().cpu()
        out["errvec0"] = obj.error().cpu()
    layer = th.TheseusLayer(opt)
    okw = dict(track_err_history=True, **kw)
    if backward:
        okw["backward_mode"] = "implicit"
    okw.update(optimizer_kwargs or {})
    if backward:
        sol, info = layer.forward(None, optimizer_kwargs=okw)
    else:
        with torch.no_grad():
            sol, info = layer.forward(None, optimizer_kwargs=okw)
    final = torch.stack([sol[f"pose_{k}"] for k in range(P)], 1)
    out.update(final=final.detach().cpu(), info=info)
    if backward:
        loss = (t(g["coef"]) * final).sum()
        loss.backward()
        out["loss"] = float(loss.detach())
        out["grads"] = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach().cpu() for k, v in leaves.items()}
    return out


def check_grads(g, grads, rel, keys=None):
    """gradients against the reference's; the radius of a shared-radius cost receives the SUM over the batch in row 0 of the
    fixture's (B, count, 1) leaf (the reference's Variable held ``leaf[:1, k]``) -- same layout on both sides."""
    for key, ref in GRAD_KEYS + ((("gnc_between", "grad_gnc_between"), ("gnc_prior", "grad_gnc_prior")) if "gnc_between" in g else ()):
        if keys is not None and key not in keys:
            continue
        want = g[ref]
        got = grads[key].double().numpy()
        # (a HingeLoss radius has no gradient: the reference records autograd noise of ~1e-23 there -- an absolute floor)
        scale = max(np.abs(want).max(), 1e-12)
        assert np.abs(got - want).max() <= rel * scale, (key, np.abs(got - want).max() / scale)
