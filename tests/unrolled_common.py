"""BackwardMode.UNROLL / TRUNCATED on an SE3 pose graph through theseus_amd's own loop (PGUnrolledIteration: thx_pg_unroll_vjp +
a copy of every differentiated iteration's factor), against the gradients the REAL reference recorded by differentiating through
its iterations (tests/golden/pg_f64_unrolled.npz, oracle/gen_golden.py:gen_pg_unrolled).  Shared by the CPU twin (stand-in
kernels) and the GPU test."""
import ast

import numpy as np
import torch


def run_pg_unrolled(th, g, tag, device, kernels=None, solver_cls=None, mode_override=None, extra_okw=None):
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    kw = ast.literal_eval(str(g[f"{tag}_kwargs"]))
    mode, iters, gn = kw.pop("mode"), kw.pop("max_iterations"), kw.pop("gauss_newton")
    if mode_override is not None:   # (a variant of the fixture's run: the caller compares)
        mode = mode_override
        if mode == "unroll":
            kw.pop("backward_num_iterations", None)
    P = int(g["P"])
    leaves = dict(meas=t(g["meas"]).requires_grad_(True), w_between=t(g["w_between"]).requires_grad_(True),
                  prior_target=t(g["prior_target"]).requires_grad_(True),
                  w_prior=t(g["w_prior"])[:, :, :1].clone().requires_grad_(True))
    obj = th.Objective(dtype=leaves["meas"].dtype)
    poses0 = t(g["poses0"])
    if f"{tag}_grad_poses0" in g:    # UNROLL: the gradient reaches the INITIAL values of the optimisation variables too
        poses0 = leaves["poses0"] = poses0.clone().requires_grad_(True)
    G = {"SE2": th.SE2, "SO3": th.SO3}.get(str(g["group"]) if "group" in g else "SE3", th.SE3)
    poses = [G(tensor=poses0[:, k] if poses0.requires_grad else poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
    # RobustCostFunction wrappers (fixtures lm_welsch_unroll / gn_huberflat_trunc): one learnable log_loss_radius
    robust = str(g[f"{tag}_robust"]) if f"{tag}_robust" in g else None
    wrap = lambda cf, nm: cf  # noqa: E731
    if robust:
        leaves["log_radius"] = t(g[f"{tag}_log_radius"]).clone().requires_grad_(True)
        radius = th.Vector(tensor=leaves["log_radius"], name="log_loss_radius")
        loss_cls = th.WelschLoss if robust.startswith("welsch") else th.HuberLoss
        wrap = lambda cf, nm: th.RobustCostFunction(cf, loss_cls, radius, name=nm, flatten_dims=robust.endswith("+flatten"))  # noqa: E731
    for k in range(g["edges"].shape[0]):
        i, j = g["edges"][k].tolist()
        obj.add(wrap(th.Between(poses[i], poses[j], G(tensor=leaves["meas"][:, k], name=f"meas_{k}"),
                                th.DiagonalCostWeight(th.Variable(leaves["w_between"][:, k], name=f"w_{k}")), name=f"between_{k}"),
                     f"robust_between_{k}"))
    for k in range(g["prior_idx"].shape[0]):
        cf = th.Difference(poses[int(g["prior_idx"][k])], G(tensor=leaves["prior_target"][:, k], name=f"prior_target_{k}"),
                           th.ScaleCostWeight(th.Variable(leaves["w_prior"][:, k], name=f"pw_{k}")), name=f"prior_{k}")
        obj.add(wrap(cf, f"robust_prior_{k}") if robust and bool(g[f"{tag}_robust_prior"]) else cf)
    lkw = dict(linearization_kwargs=dict(kernels=kernels)) if kernels is not None else {}
    cls = th.GaussNewton if gn else th.LevenbergMarquardt
    tol = float(g[f"{tag}_rel_tol"]) if f"{tag}_rel_tol" in g else 0.0
    step = float(g[f"{tag}_step"]) if f"{tag}_step" in g else 1.0
    if solver_cls is not None:
        lkw["linear_solver_cls"] = solver_cls
    opt = cls(obj, max_iterations=iters, step_size=step, abs_err_tolerance=0.0, rel_err_tolerance=tol, **lkw)
    sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(backward_mode=mode, track_err_history=True, **kw,
                                                                         **(extra_okw or {})))
    final = torch.stack([sol[f"pose_{k}"] for k in range(P)], 1)
    loss = (t(g["coef"]) * final).sum()
    loss.backward()
    if mode_override is not None or extra_okw:
        return info, final.detach(), leaves["meas"].grad
    # (the Welsch case is still descending after its 4 iterations -- error 5.2 -> 0.9 -- and its down-weighted system is less
    #  well conditioned: the tiled Cholesky and LAPACK's differ by 2e-9 on the final poses there, 1e-10 elsewhere)
    tol_x = 2e-8 if robust else 1e-9
    np.testing.assert_allclose(final.detach().cpu().numpy(), g[f"{tag}_final"], rtol=0, atol=tol_x)
    np.testing.assert_allclose(info.err_history.numpy(), g[f"{tag}_err_history"], rtol=1e-6)
    assert abs(float(loss.detach()) - float(g[f"{tag}_loss"])) < 10 * tol_x
    for key in ("meas", "w_between", "prior_target", "w_prior") + (("log_radius",) if robust else ()):
        got, want = leaves[key].grad.cpu().numpy(), g[f"{tag}_grad_{key}"]
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-6 * np.abs(want).max(), err_msg=key)
    if "poses0" in leaves:
        # (raw-entry gradients of the start: dominated by the directions OFF the manifold, amplified by the weak gauge prior --
        #  the oracle's own autograd differs from the reference's by 1e-6 of the largest entry on the worst problem, 1e-13 on the best)
        got, want = leaves["poses0"].grad.cpu().numpy(), g[f"{tag}_grad_poses0"]
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-5 * np.abs(want).max(), err_msg="poses0")
    if tol:   # problems converge -- and are frozen -- at different differentiated iterations; the loop stops early
        assert info.converged_iter.tolist() == g[f"{tag}_conv"].tolist()
        assert [int(s_.value) for s_ in info.status] == g[f"{tag}_status"].tolist()
    return info
