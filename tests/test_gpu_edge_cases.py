"""-m gpu: edge cases of the packed problem layouts against the oracle: no priors, a single edge, batch 1, batch-shared
vs batched auxiliary tensors, a pose touched by one cost only, bundle adjustment without point priors."""
import dataclasses

import numpy as np
import pytest
import torch

from theseus_amd._lib import THX_ERR_CHUNKS

from oracle import pose_graph as opg
from tests.helpers import golden_problem, load_golden

pytestmark = pytest.mark.gpu


def _check_pg(p, poses0, rel=5e-12):
    from tests.gpu_helpers import alloc_dense, sym_from_lower, to_device_problem
    from theseus_amd.kernels import default_kernels
    K = default_kernels()
    s, t = to_device_problem(p, poses0)
    ds = s.on("cuda")
    B, n = poses0.shape[0], s.num_cols
    H, gv, _ = alloc_dense(B, n, poses0.dtype)
    K.pg_assemble(ds, t, H, gv)
    A, b = opg.dense_linearize(p, poses0)
    AtA, Atb = opg.hessian(A, b)
    assert (sym_from_lower(H, n).cpu() - AtA).abs().max() <= rel * AtA.abs().max()
    assert (gv.cpu() - Atb[..., 0]).abs().max() <= rel * max(Atb.abs().max().item(), 1e-300)
    part = torch.empty(THX_ERR_CHUNKS, B, dtype=poses0.dtype, device="cuda")
    err = torch.empty(B, dtype=poses0.dtype, device="cuda")
    K.pg_error(ds, t, part, err)
    np.testing.assert_allclose(err.cpu().numpy(), opg.error_metric(p, poses0).numpy(), rtol=1e-12)


def test_pose_graph_without_priors_single_edge_batch_one():
    g = load_golden("pg_f64_lm")
    p, poses0, _ = golden_problem(g)
    none = dataclasses.replace(p, prior_idx=p.prior_idx[:0], prior_target=p.prior_target[:, :0], w_prior=p.w_prior[:, :0],
                               cost_order=None)
    _check_pg(none, poses0)                                    # K = 0
    one = dataclasses.replace(none, edges=p.edges[:1], meas=p.meas[:, :1], w_between=p.w_between[:, :1], cost_order=None)
    _check_pg(one, poses0)                                     # E = 1: most poses have no cost at all
    b1 = dataclasses.replace(p, meas=p.meas[:1], prior_target=p.prior_target[:1])
    _check_pg(b1, poses0[:1])                                  # B = 1


def test_pose_graph_shared_and_batched_auxiliaries_agree():
    g = load_golden("pg_f64_lm_adaptive_ellips")              # batched weights
    p, poses0, _ = golden_problem(g)
    B = poses0.shape[0]
    shared = dataclasses.replace(p, meas=p.meas[:1].expand(1, -1, -1, -1).contiguous(), w_between=p.w_between[:1].contiguous(),
                                 prior_target=p.prior_target[:1].contiguous())
    _check_pg(shared, poses0)                                   # every auxiliary with batch stride 0
    full = dataclasses.replace(shared, meas=shared.meas.expand(B, -1, -1, -1).contiguous(),
                               w_between=shared.w_between.expand(B, -1, -1).contiguous(),
                               prior_target=shared.prior_target.expand(B, -1, -1, -1).contiguous(),
                               w_prior=shared.w_prior.expand(B, -1, -1).contiguous())
    _check_pg(full, poses0)                                     # the same values, all batched


def test_lm_on_a_graph_with_an_isolated_pose_fails_like_the_reference():
    """A pose that no cost touches makes H singular: Gauss-Newton must report FAIL (linalg.cholesky raises in the
    reference, nonlinear_least_squares.py:138-152), LM with damping must run."""
    import warnings
    import theseus_amd as th
    g = load_golden("pg_f64_gn")
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    obj = th.Objective(dtype=torch.float64)
    poses = [th.SE3(tensor=t(g["poses0"])[:, k].clone(), name=f"pose_{k}") for k in range(3)]
    w = th.DiagonalCostWeight(th.Variable(t(g["w_between"])[:, 0].clone(), name="w"))
    obj.add(th.Between(poses[0], poses[1], th.SE3(tensor=t(g["meas"])[:, 0].clone(), name="m"), w, name="e01"))
    obj.add(th.Difference(poses[0], th.SE3(tensor=t(g["prior_target"])[:, 0].clone(), name="t"),
                          th.ScaleCostWeight(torch.tensor(1.0, dtype=torch.float64, device="cuda")), name="prior"))
    # pose_2 is registered through a zero-information edge only
    w0 = th.DiagonalCostWeight(th.Variable(torch.zeros(1, 6, dtype=torch.float64, device="cuda"), name="w0"))
    obj.add(th.Between(poses[1], poses[2], th.SE3(tensor=t(g["meas"])[:, 1].clone(), name="m2"), w0, name="e12"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        info = th.GaussNewton(obj, max_iterations=2).optimize()
    assert all(s == th.NonlinearOptimizerStatus.FAIL for s in info.status)
    obj.update({f"pose_{k}": t(g["poses0"])[:, k].clone() for k in range(3)})
    info = th.LevenbergMarquardt(obj, max_iterations=3).optimize(damping=1e-3)
    assert all(s != th.NonlinearOptimizerStatus.FAIL for s in info.status)


def test_bundle_adjustment_without_point_priors():
    import theseus_amd as th
    from tests.ba_common import build_ba_objective
    from tests.helpers import ba_problem
    g = dict(load_golden("ba_f64_lm"))
    keep = g["cost_kind"] != 2                                  # drop every Point3 regulariser
    g["cost_kind"], g["cost_idx"] = g["cost_kind"][keep], g["cost_idx"][keep]
    obj, _, _ = build_ba_objective(th, g, "cuda")
    opt = th.LevenbergMarquardt(obj, max_iterations=1)
    lin, solver = opt.linear_solver.linearization, opt.linear_solver
    obj.update()
    lin.linearize()
    delta = solver.solve(damping=torch.full((g["cams0"].shape[0],), 1e-2, dtype=torch.float64, device="cuda"),
                         ellipsoidal_damping=True, damping_eps=1e-8)
    p, state0, _, used = ba_problem(g)
    p = dataclasses.replace(p, pt_prior_idx=p.pt_prior_idx[:0], pt_prior_target=p.pt_prior_target[:, :0],
                            w_pt_prior=p.w_pt_prior[:, :0], cost_order=[c for c in p.cost_order if c[0] != "pt_prior"],
                            var_order=[("cam", i) for i in range(p.num_cams)] + [("pt", i) for i in range(p.num_points)])
    # theseus_amd orders the points by first appearance among the observations: re-index the oracle problem accordingly
    order = []
    for q in p.obs_pt.tolist():
        if q not in order:
            order.append(q)
    inv = {q: k for k, q in enumerate(order)}
    p = dataclasses.replace(p, obs_pt=torch.tensor([inv[q] for q in p.obs_pt.tolist()]))
    state0 = (state0[0], state0[1][:, order])
    A, b = p.dense_linearize(state0)
    AtA, Atb = opg.hessian(A, b)
    want = opg.solve(AtA, Atb, 1e-2 * torch.ones(AtA.shape[0], dtype=torch.float64), True, 1e-8)
    assert (delta.cpu() - want).abs().max() <= 1e-8 * max(1.0, want.abs().max().item())
