"""Implicit backward mode on CPU: theseus_amd's host logic (nonlinear.py implicit branch, autograd.ImplicitStep,
gradient reductions over broadcast dimensions) with the TEST stand-in kernels, against gradients recorded from the
real reference through th.TheseusLayer(backward_mode="implicit") (oracle/gen_golden.py:gen_implicit)."""
import pytest
import torch

from tests.helpers import load_golden
from tests.helpers import golden_problem  # noqa: F401
from tests.implicit_common import check_against_reference, run_implicit


@pytest.mark.parametrize("name", ["pg_f64_implicit", "pg_f64_implicit_b", "pg2_f64_implicit", "pg3_f64_implicit"])
def test_implicit_gradients_match_reference(name):
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    g = load_golden(name)
    final, loss, grads, info, _, _ = run_implicit(th, g, "cpu", OracleKernels())
    check_against_reference(g, final, loss, grads)
    assert info.iters_done == golden_problem(g)[2]["max_iterations"]


def test_stale_factor_is_detected_and_unroll_with_grad_refused():
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    g = load_golden("pg_f64_implicit")
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    meas = t(g["meas"]).requires_grad_(True)
    obj = th.Objective(dtype=torch.float64)
    poses = [th.SE3(tensor=t(g["poses0"])[:, k].clone(), name=f"pose_{k}") for k in range(int(g["P"]))]
    w = th.DiagonalCostWeight(th.Variable(t(g["w_between"])[:, 0], name="w"))
    for k in range(g["edges"].shape[0]):
        i, j = g["edges"][k].tolist()
        obj.add(th.Between(poses[i], poses[j], th.SE3(tensor=meas[:, k], name=f"meas_{k}"), w, name=f"b_{k}"))
    obj.add(th.Difference(poses[0], th.SE3(tensor=t(g["prior_target"])[:, 0], name="tgt"),
                          th.ScaleCostWeight(torch.tensor(1.0, dtype=torch.float64)), name="prior"))
    opt = th.LevenbergMarquardt(obj, linearization_kwargs=dict(kernels=OracleKernels()), max_iterations=3)
    layer = th.TheseusLayer(opt)
    # (unrolled differentiation of an SE3 pose graph is fused since round 4: tests/test_unrolled_host.py; a trust-region optimizer
    #  with unrolled gradients is what is still refused)
    dog = th.Dogleg(obj, linearization_kwargs=dict(kernels=OracleKernels()), max_iterations=3)
    with pytest.raises(NotImplementedError, match="trust-region"):
        th.TheseusLayer(dog).forward(None, optimizer_kwargs=dict(backward_mode="unroll"))
    with torch.no_grad():  # without gradients the default mode is the plain loop
        th.TheseusLayer(dog).forward(None, optimizer_kwargs=dict(backward_mode="unroll"))
    sol, _ = layer.forward(None, optimizer_kwargs=dict(backward_mode="implicit"))
    loss = sum(v.sum() for v in sol.values())
    with torch.no_grad():
        layer.forward(None)                      # a later factorisation on the same optimizer ...
    with pytest.raises(RuntimeError, match="overwritten"):
        loss.backward()                          # ... invalidates the cached factor of the first graph
