"""Bundle-adjustment fixtures (tests/golden/ba_*.npz, oracle/gen_golden.py:gen_ba) -> theseus_amd objective, built in the
reference's order so that the reference column layout (insertion order) can be recovered for comparisons."""
import numpy as np
import torch

from tests.helpers import ba_problem


def build_ba_objective(th, g, device="cpu"):
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    dtype = torch.from_numpy(g["cams0"]).dtype
    C, Np, O = int(g["C"]), int(g["Np"]), g["obs_cam"].shape[0]
    cams0, pts0, feat = t(g["cams0"]), t(g["pts0"]), t(g["feat"])
    obj = th.Objective(dtype=dtype)
    cam_v = [th.SE3(tensor=cams0[:, i].clone(), name=f"Cam{i}") for i in range(C)]
    pt_v = [th.Point3(tensor=pts0[:, i].clone(), name=f"Pt{i}") for i in range(Np)]
    fl = [th.Vector(tensor=t(g["focal"])[:, i].clone(), name=f"fl{i}") for i in range(C)]
    k1 = [th.Vector(tensor=t(g["k1"])[:, i].clone(), name=f"k1_{i}") for i in range(C)]
    k2 = [th.Vector(tensor=t(g["k2"])[:, i].clone(), name=f"k2_{i}") for i in range(C)]
    w = th.ScaleCostWeight(torch.tensor(1.0, dtype=dtype, device=device))
    robust = str(g["robust"])
    radius = th.Vector(tensor=torch.tensor([[float(g["log_radius"])]], dtype=dtype, device=device), name="log_loss_radius")
    for o in range(O):
        c, p = int(g["obs_cam"][o]), int(g["obs_pt"][o])
        cf = th.Reprojection(camera_pose=cam_v[c], world_point=pt_v[p], focal_length=fl[c], calib_k1=k1[c], calib_k2=k2[c],
                             image_feature_point=th.Point2(tensor=feat[:, o].clone(), name=f"Feat{o}"), weight=w, name=f"reproj_{o}")
        if robust:
            cf = th.RobustCostFunction(cf, th.HuberLoss if robust == "huber" else th.WelschLoss, radius, name=f"robust_{o}")
        obj.add(cf)
    # priors in the fixture's cost order (regularisers interleaved by variable insertion order, then the strong camera priors)
    tgt_c, w_c = t(g["cam_prior_target"]), t(g["w_cam_prior"])
    w_p = t(g["w_pt_prior"])
    zero_pt = th.Point3(tensor=torch.zeros(1, 3, dtype=dtype, device=device), name="zero_point")
    seen = {}
    for kind, k in zip(g["cost_kind"].tolist(), g["cost_idx"].tolist()):
        if kind == 1:
            i = int(g["cam_prior_idx"][k])
            key = tgt_c[:, k].cpu().numpy().tobytes()
            target = seen.setdefault(key, th.SE3(tensor=tgt_c[:, k].clone(), name=f"cam_target_{k}"))
            obj.add(th.Difference(cam_v[i], target, th.ScaleCostWeight(w_c[:, k, :1].clone()), name=f"cam_prior_{k}"))
        elif kind == 2:
            i = int(g["pt_prior_idx"][k])
            obj.add(th.Difference(pt_v[i], zero_pt, th.ScaleCostWeight(w_p[:, k, :1].clone()), name=f"pt_prior_{k}"))
        elif kind == 3:   # camera-camera Between (odometry): tests/golden/ba_f64_camcam_lm.npz
            i, j = g["cc_edges"][k].tolist()
            obj.add(th.Between(cam_v[i], cam_v[j], th.SE3(tensor=t(g["cc_meas"])[:, k].clone(), name=f"odo_{k}"),
                               th.DiagonalCostWeight(th.Variable(t(g["w_cc"])[:, k].clone(), name=f"w_odo_{k}")), name=f"odometry_{k}"))
    return obj, cam_v, pt_v


def reference_columns(g):
    """Index vector ``cols`` with delta_reference[:, j] = delta_internal[:, cols[j]] (internal order: cameras, then the
    objective's points in insertion order = order of first appearance in the fixture's variable list)."""
    kinds, idxs = g["var_kind"].tolist(), g["var_idx"].tolist()
    C = int(g["C"])
    pts_in_order = [i for k, i in zip(kinds, idxs) if k == 1]
    # theseus_amd's objective registers variables in the same insertion order as the reference's, so the internal point
    # order is the order in which points first appear among the observations
    seen, order = set(), []
    for p in g["obs_pt"].tolist():
        if p not in seen:
            seen.add(p)
            order.append(p)
    pos = {p: k for k, p in enumerate(order)}
    cols = []
    for k, i in zip(kinds, idxs):
        cols += list(range(6 * i, 6 * i + 6)) if k == 0 else [6 * C + 3 * pos[i] + a for a in range(3)]
    assert sorted(pts_in_order) == sorted(order)
    return np.array(cols), order


def run_ba(th, g, kernels=None, device="cpu"):
    from tests.helpers import ba_problem  # noqa: F401
    obj, cam_v, pt_v = build_ba_objective(th, g, device)
    import ast
    kw = ast.literal_eval(str(g["opt_kwargs"]))
    gn = kw.pop("gauss_newton")
    okw = dict(max_iterations=kw.pop("max_iterations"), step_size=kw.pop("step_size"), abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    if kernels is not None:
        okw["linearization_kwargs"] = dict(kernels=kernels)
    opt = (th.GaussNewton if gn else th.LevenbergMarquardt)(obj, **okw)
    assert type(opt.linear_solver).__name__ == "HipSchurSolver"
    deltas = []
    sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(
        track_err_history=True, end_iter_callback=lambda o, i, d, it: deltas.append(d.clone()), **kw))
    cams = torch.stack([sol[f"Cam{i}"] for i in range(int(g["C"]))], 1)
    used = sorted(set(g["obs_pt"].tolist()))
    pts = torch.stack([sol[f"Pt{i}"] for i in used], 1)
    return cams, pts, used, deltas, info, opt


def run_ba_implicit(th, g, kernels=None, device="cpu", opt_kwargs=None):
    """The objective of oracle/gen_golden.py:gen_ba_implicit on theseus_amd's classes with the differentiable leaves of that
    fixture; LM + backward_mode="implicit"; returns the solution, the loss and the gradients as numpy arrays."""
    import ast
    t = lambda a: torch.from_numpy(np.asarray(a)).to(device)  # noqa: E731
    dtype = torch.float64
    C, Np, O = int(g["C"]), int(g["Np"]), g["obs_cam"].shape[0]
    n_reg = int(g["n_reg_cam"])
    leaves = dict(feat=t(g["feat"]), focal=t(g["focal"]), k1=t(g["k1"]), k2=t(g["k2"]),
                  log_radius=torch.tensor([[float(g["log_radius"])]], dtype=dtype, device=device),
                  w_obs=torch.tensor(1.0, dtype=dtype, device=device), gt_cams=t(g["cam_prior_target"])[:, n_reg:].clone(),
                  w_strong=100 * torch.ones(1, dtype=dtype, device=device),
                  w_reg=t(g["w_cam_prior"])[0, 0, :1].clone())
    leaves = {k: v.clone().requires_grad_(True) for k, v in leaves.items()}
    cams0, pts0 = t(g["cams0"]), t(g["pts0"])
    obj = th.Objective(dtype=dtype)
    if "grad_cams0" in g:   # UNROLL fixtures: the gradient reaches the INITIAL cameras / points too
        leaves["cams0"], leaves["pts0"] = cams0.clone().requires_grad_(True), pts0.clone().requires_grad_(True)
        cam_v = [th.SE3(tensor=leaves["cams0"][:, i], name=f"Cam{i}") for i in range(C)]
        pt_v = [th.Point3(tensor=leaves["pts0"][:, i], name=f"Pt{i}") for i in range(Np)]
    else:
        cam_v = [th.SE3(tensor=cams0[:, i].clone(), name=f"Cam{i}") for i in range(C)]
        pt_v = [th.Point3(tensor=pts0[:, i].clone(), name=f"Pt{i}") for i in range(Np)]
    fl = [th.Vector(tensor=leaves["focal"][:, i], name=f"fl{i}") for i in range(C)]
    k1 = [th.Vector(tensor=leaves["k1"][:, i], name=f"k1_{i}") for i in range(C)]
    k2 = [th.Vector(tensor=leaves["k2"][:, i], name=f"k2_{i}") for i in range(C)]
    w = th.ScaleCostWeight(th.Variable(leaves["w_obs"].view(1, 1), name="w_obs"))
    radius = th.Vector(tensor=leaves["log_radius"], name="log_loss_radius")
    for o in range(O):
        c, p = int(g["obs_cam"][o]), int(g["obs_pt"][o])
        cf = th.Reprojection(camera_pose=cam_v[c], world_point=pt_v[p], focal_length=fl[c], calib_k1=k1[c], calib_k2=k2[c],
                             image_feature_point=th.Point2(tensor=leaves["feat"][:, o], name=f"Feat{o}"), weight=w, name=f"reproj_{o}")
        obj.add(th.RobustCostFunction(cf, th.HuberLoss, radius, name=f"robust_{o}", flatten_dims=str(g["robust"]).endswith("+flatten")))
    dw = th.ScaleCostWeight(th.Variable(leaves["w_reg"].view(1, 1), name="w_reg"))
    ident = th.SE3(tensor=torch.eye(3, 4, dtype=dtype, device=device).unsqueeze(0), name="zero_se3")
    zero_pt = th.Point3(tensor=torch.zeros(1, 3, dtype=dtype, device=device), name="zero_point")
    for kind, k in zip(g["cost_kind"].tolist(), g["cost_idx"].tolist()):   # regularisers in the fixture's cost order
        if kind == 1 and k < n_reg:
            obj.add(th.Difference(cam_v[int(g["cam_prior_idx"][k])], ident, dw, name=f"reg_cam_{k}"))
        elif kind == 2:
            obj.add(th.Difference(pt_v[int(g["pt_prior_idx"][k])], zero_pt, dw, name=f"reg_pt_{k}"))
    cw = th.ScaleCostWeight(th.Variable(leaves["w_strong"].view(1, 1), name="w_strong"))
    for k, i in enumerate((0, C - 1)):
        obj.add(th.Difference(cam_v[i], th.SE3(tensor=leaves["gt_cams"][:, k], name=f"gt_cam{i}"), cw, name=f"camera_diff_{i}"))
    if "cc_edges" in g:   # camera-camera Between (odometry) costs with differentiable measurements / weights
        leaves["cc_meas"] = t(g["cc_meas"]).clone().requires_grad_(True)
        leaves["w_cc"] = t(g["w_cc"]).clone().requires_grad_(True)
        for k, (i, j) in enumerate(g["cc_edges"].tolist()):
            obj.add(th.Between(cam_v[i], cam_v[j], th.SE3(tensor=leaves["cc_meas"][:, k], name=f"odo_{k}"),
                               th.DiagonalCostWeight(th.Variable(leaves["w_cc"][:, k], name=f"w_odo_{k}")), name=f"odometry_{k}"))
    kw = ast.literal_eval(str(g["opt_kwargs"]))
    kw.pop("gauss_newton")
    mode = kw.pop("backward_mode", "implicit")     # ("unroll" / "truncated": the ba_f64_*unroll* / *trunc* fixtures)
    okw = dict(max_iterations=kw.pop("max_iterations"), step_size=kw.pop("step_size"), abs_err_tolerance=0.0,
               rel_err_tolerance=float(g["rel_tol"]) if "rel_tol" in g else 0.0)   # (ba_f64_trunc_conv_lm: convergence tests on)
    if kernels is not None:
        okw["linearization_kwargs"] = dict(kernels=kernels)
    okw.update(opt_kwargs or {})
    opt = th.LevenbergMarquardt(obj, **okw)
    layer = th.TheseusLayer(opt)
    if device != "cpu" and opt_kwargs is not None:   # (the REAL theseus keeps Objective.device separately from its tensors')
        layer.to(device)
    sol, info = layer.forward(None, optimizer_kwargs=dict(backward_mode=mode, track_err_history=True, **kw))
    used = sorted(set(g["obs_pt"].tolist()))
    final_c = torch.stack([sol[f"Cam{i}"] for i in range(C)], 1)
    final_p = torch.stack([sol[f"Pt{i}"] for i in used], 1)
    loss = (t(g["coef_c"]) * final_c).sum() + (t(g["coef_p"]) * final_p).sum()
    loss.backward()
    out = dict(final_cams=final_c.detach().cpu().numpy(), final_pts=final_p.detach().cpu().numpy(), loss=float(loss.detach()),
               err_history=info.err_history.cpu().numpy(), converged_iter=info.converged_iter.cpu().numpy(),
               status=np.array([int(s_.value) for s_ in info.status]))
    for k, v in leaves.items():
        out["grad_" + k] = v.grad.detach().cpu().numpy()
    return out
