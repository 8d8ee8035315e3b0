"""RobustCostFunction (Welsch / Huber) through the host path on CPU: theseus_amd's mirror classes, packer, LM loop and
implicit backward with the TEST stand-in kernels (tests/oracle_kernels.py) reproduce the reference's PUBLISHED
known-answer test (tests/theseus_tests/test_pgo_benchmark.py:34-39).  The GPU twin is tests/test_gpu_robust.py."""
import pytest
import torch

from tests.pgo_kat_common import kat, outer_loop


def build_kat_objective(th, g, sl, log_radius, device="cpu"):
    """examples/pose_graph/pose_graph_synthetic.py:121-176 on the fixture's batch ``sl``."""
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    P = g["poses0"].shape[1]
    obj = th.Objective(dtype=torch.float64)
    poses0 = t(g["poses0"][sl])
    poses = [th.SE3(tensor=poses0[:, k].clone(), name=f"VERTEX_SE3__{k}") for k in range(P)]
    radius = th.Vector(tensor=log_radius, name="log_loss_radius")
    for k, (i, j) in enumerate(g["edges"].tolist()):
        w = th.DiagonalCostWeight(th.Variable(t(g["w_between"])[:, k].clone(), name=f"EDGE_WEIGHT__{k}"))
        m = th.SE3(tensor=t(g["meas"][sl])[:, k].clone(), name=f"EDGE_SE3__{k}")
        obj.add(th.RobustCostFunction(th.Between(poses[i], poses[j], m, w, name=f"between_{k}"), th.WelschLoss, radius,
                                      name=f"robust_between_{k}"))
    reg = th.ScaleCostWeight(torch.tensor(float(g["reg_w"]), dtype=torch.float64, device=device))
    obj.add(th.Difference(poses[0], th.SE3(tensor=poses0[:, 0].clone(), name="VERTEX_SE3__0__PRIOR"), reg, name="prior"))
    known = th.ScaleCostWeight(torch.tensor(float(g["known_w"]), dtype=torch.float64, device=device))
    for i in g["gt_idx"].tolist():
        obj.add(th.Difference(poses[i], th.SE3(tensor=t(g["gt"][sl])[:, i].clone(), name=f"VERTEX_SE3_GT__{i}"), known,
                              name=f"pose_diff_{i}"))
    return obj, poses


def run_kat(th, kernels=None, device="cpu"):
    g = kat()

    def inner(sl, log_radius):
        obj, _ = build_kat_objective(th, g, sl, log_radius.to(device), device)
        kw = dict(linearization_kwargs=dict(kernels=kernels)) if kernels is not None else {}
        opt = th.LevenbergMarquardt(obj, max_iterations=int(g["max_iters"]), step_size=float(g["step_size"]),
                                    linear_solver_cls=th.HipCholeskySolver, **kw)
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(
            backward_mode="implicit", track_err_history=True, adaptive_damping=True, **{"__keep_final_step_size__": True}))
        return torch.stack([sol[f"VERTEX_SE3__{k}"] for k in range(g["poses0"].shape[1])], 1)
    return outer_loop(g, inner), g["losses_published"]


def test_reference_pgo_known_answer_through_host_path():
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    losses, want = run_kat(th, OracleKernels())
    for a, b in zip(losses, want):
        assert a == pytest.approx(b, rel=1e-10, abs=1e-10), (losses, want)


@pytest.mark.parametrize("name", ["pg_f64_mixed_robust", "pg2_f64_mixed_robust", "pg_f64_mixed_hinge", "pg_f64_mixed_gnc"])
def test_mixed_and_flattened_robust_costs_through_host_path(name):
    """Plain, Welsch, Huber and flatten_dims=True costs mixed inside one role (robust_cost_function.py:52-135): the packer's
    per-cost loss table (theseus_amd/packed.py), Objective.error(), the LM loop and the implicit backward incl. the gradient of
    every log_loss_radius, against the REAL reference's fixture.  GPU twin: tests/test_gpu_robust.py."""
    import numpy as np
    import theseus_amd as th
    from tests.helpers import load_golden
    from tests.mixed_robust_common import check_grads, run_mixed_implicit
    from tests.oracle_kernels import OracleKernels
    g = load_golden(name)
    r = run_mixed_implicit(th, g, "cpu", OracleKernels())
    packed = r["opt"].linear_solver.linearization.packed
    assert packed.loss_between is not None and packed.loss_prior is not None      # both roles are mixed
    if name.endswith("hinge"):     # plain, Huber, Hinge and flattened Hinge; the first prior a flattened Hinge cost
        assert sorted(set(packed.loss_between.tolist())) == [0, 2, 3, 7] and packed.loss_prior.tolist()[0] == 7
    elif name.endswith("gnc"):     # plain, Huber, Geman-McClure (GNCRobustCostFunction) plain and flattened; the first prior Geman-McClure
        assert sorted(set(packed.loss_between.tolist())) == [0, 2, 8, 12] and packed.loss_prior.tolist()[0] == 8
    else:
        assert sorted(set(packed.loss_between.tolist())) == [0, 1, 2, 5, 6] and packed.loss_prior.tolist()[0] == 5
    np.testing.assert_allclose(r["err0"].numpy(), g["err0"], rtol=1e-12)
    np.testing.assert_allclose(r["errvec0"].numpy(), g["errvec0"], rtol=0, atol=1e-12 * np.abs(g["errvec0"]).max())
    np.testing.assert_allclose(r["info"].err_history.numpy(), g["err_history"], rtol=1e-6)
    np.testing.assert_allclose(r["final"].numpy(), g["final"], rtol=0, atol=5e-8)
    assert abs(r["loss"] - float(g["loss"])) < 1e-6
    check_grads(g, r["grads"], 2e-6)
