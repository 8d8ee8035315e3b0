"""Generic objectives on theseus_amd's own API without a GPU (theseus_amd/euclidean.py with the TEST stand-in kernels):
BASELINE.json configs[0] against the REAL reference's run.  GPU twin: tests/test_gpu_generic.py."""
import numpy as np
import pytest
import torch

from tests.helpers import load_golden
from tests.simple_example_common import check_simple_example, run_simple_example


def test_simple_example_config0_matches_the_reference():
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    g = load_golden("simple_example")
    r = run_simple_example(th, g, "cpu", OracleKernels())
    lin = r["opt"].linear_solver.linearization
    assert type(lin.packed).__name__ == "PackedEuclidean" and lin.n == 1 and r["opt2"].linear_solver.linearization.n == 2
    check_simple_example(g, r)


def test_generic_linearization_matches_torch_on_a_two_cost_objective():
    """Two AutoDiff costs sharing one of two variables, a second call after ``Objective.update`` (re-pack), ``A`` / ``b`` / ``AtA``
    / ``Atb`` / ``Av`` / ``error_metric`` against plain torch on the same formulas; a non-Euclidean optimisation variable is
    refused."""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    dt, B = torch.float64, 5
    gen = torch.Generator().manual_seed(3)
    u, w = th.Vector(2, name="u", dtype=dt), th.Vector(3, name="w", dtype=dt)
    d1 = th.Variable(torch.randn(B, 4, dtype=dt, generator=gen), name="d1")
    d2 = th.Variable(torch.randn(1, 3, dtype=dt, generator=gen), name="d2")

    def f1(optim_vars, aux_vars):    # (B, 4): mixes u and w
        uu, ww = optim_vars[0].tensor, optim_vars[1].tensor
        return torch.cat([uu * ww[:, :2], torch.sin(uu) + ww[:, 2:]], 1) - aux_vars[0].tensor

    def f2(optim_vars, aux_vars):    # (B, 3): w only
        return optim_vars[0].tensor ** 2 - aux_vars[0].tensor
    obj = th.Objective(dtype=dt)
    obj.add(th.AutoDiffCostFunction([u, w], f1, 4, aux_vars=[d1], cost_weight=th.ScaleCostWeight(torch.tensor(2.0, dtype=dt)), name="c1"))
    obj.add(th.AutoDiffCostFunction([w], f2, 3, aux_vars=[d2], cost_weight=th.DiagonalCostWeight(torch.tensor([[1.0, 0.5, 2.0]], dtype=dt)), name="c2"))
    lin = th.HipLinearization(obj, kernels=OracleKernels())
    for trial in range(2):
        U, W = torch.randn(B, 2, dtype=dt, generator=gen), torch.randn(B, 3, dtype=dt, generator=gen)
        obj.update({"u": U, "w": W})
        lin.linearize()
        # plain torch
        X = torch.cat([U, W], 1).requires_grad_(True)

        def err(x):
            uu, ww = x[:, :2], x[:, 2:]
            e1 = (torch.cat([uu * ww[:, :2], torch.sin(uu) + ww[:, 2:]], 1) - d1.tensor) * 2.0
            e2 = (ww ** 2 - d2.tensor) * torch.tensor([[1.0, 0.5, 2.0]], dtype=dt)
            return torch.cat([e1, e2], 1)
        e = err(X)
        A = torch.stack([torch.autograd.grad(e[:, r].sum(), X, retain_graph=True)[0] for r in range(7)], 1)   # (B, 7, 5)
        np.testing.assert_allclose(lin.A.numpy(), A.numpy(), atol=1e-13)
        np.testing.assert_allclose(lin.b.numpy(), -e.detach().numpy(), atol=1e-13)
        np.testing.assert_allclose(lin.AtA.numpy(), (A.transpose(1, 2) @ A).numpy(), atol=1e-12)
        np.testing.assert_allclose(lin.Atb.squeeze(2).numpy(), -(A.transpose(1, 2) @ e.detach().unsqueeze(2)).squeeze(2).numpy(), atol=1e-12)
        vv = torch.randn(B, 5, dtype=dt, generator=gen)
        np.testing.assert_allclose(lin.Av(vv).numpy(), (A @ vv.unsqueeze(2)).squeeze(2).numpy(), atol=1e-12)
        np.testing.assert_allclose(obj.error_metric().numpy(), 0.5 * (e.detach() ** 2).sum(1).numpy(), rtol=1e-13)
        np.testing.assert_allclose(obj.error().numpy(), e.detach().numpy(), atol=1e-13)
    pose = th.SE3(tensor=torch.eye(3, 4, dtype=dt).unsqueeze(0), name="pose")
    bad = th.Objective(dtype=dt)
    bad.add(th.AutoDiffCostFunction([pose], lambda optim_vars, aux_vars: optim_vars[0].tensor.reshape(-1, 12), 12))
    with pytest.raises(th.UnsupportedObjective):
        th.HipLinearization(bad, kernels=OracleKernels())


@pytest.mark.parametrize("tag,cls,okw", [
    ("gn", "GaussNewton", dict(track_best_solution=True, track_state_history=True)),
    ("lm_tiny", "LevenbergMarquardt", dict(damping=1e-6, adaptive_damping=True, track_best_solution=True)),
    ("dogleg", "Dogleg", dict(track_state_history=True)),
    ("lm_ellips", "LevenbergMarquardt", dict(damping=0.1, ellipsoidal_damping=True, adaptive_damping=True))])
def test_generic_path_optimizer_variants_match_the_reference(tag, cls, okw):
    """The two-variable fit y = a exp(b x) through Gauss-Newton / adaptive LM (plain and ellipsoidal damping) / Dogleg on the
    generic path, with track_best_solution and track_state_history, against the REAL reference's runs
    (tests/golden/simple_example.npz: v_* entries)."""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    g = load_golden("simple_example")
    dt = torch.float64
    xs, ys = torch.from_numpy(g["v_x"]), torch.from_numpy(g["v_y"])
    B, N = xs.shape
    a, b = th.Vector(1, name="a", dtype=dt), th.Vector(1, name="b", dtype=dt)
    x, y = th.Variable(xs.clone(), name="x"), th.Variable(ys.clone(), name="y")

    def f(optim_vars, aux_vars):
        return aux_vars[1].tensor - optim_vars[0].tensor * torch.exp(optim_vars[1].tensor * aux_vars[0].tensor)
    obj = th.Objective(dtype=dt)
    obj.add(th.AutoDiffCostFunction([a, b], f, N, aux_vars=[x, y], cost_weight=th.ScaleCostWeight(torch.tensor(1.0, dtype=dt))))
    opt = getattr(th, cls)(obj, max_iterations=8, abs_err_tolerance=0.0, rel_err_tolerance=0.0,
                           linearization_kwargs=dict(kernels=OracleKernels()))
    with torch.no_grad():
        sol, info = th.TheseusLayer(opt).forward({"a": torch.ones(B, 1, dtype=dt), "b": 2.5 * torch.ones(B, 1, dtype=dt)},
                                                 optimizer_kwargs=dict(track_err_history=True, **okw))
    np.testing.assert_allclose(info.err_history.numpy(), g[f"v_{tag}_err"], rtol=1e-6)
    np.testing.assert_allclose(sol["a"].numpy(), g[f"v_{tag}_a"], rtol=1e-9)
    np.testing.assert_allclose(sol["b"].numpy(), g[f"v_{tag}_b"], rtol=1e-9)
    if okw.get("track_best_solution"):
        np.testing.assert_allclose(info.best_solution["a"].numpy(), g[f"v_{tag}_best_a"], rtol=1e-9)
        np.testing.assert_allclose(info.best_solution["b"].numpy(), g[f"v_{tag}_best_b"], rtol=1e-9)
        np.testing.assert_allclose(info.best_err.numpy(), g[f"v_{tag}_best_err"], rtol=1e-9)
    if okw.get("track_state_history"):
        np.testing.assert_allclose(info.state_history["b"].numpy(), g[f"v_{tag}_hist_b"], rtol=1e-9)


@pytest.mark.parametrize("tag", ["gn_unroll", "gn_trunc", "lm_unroll", "lm_trunc", "gn_trunc_conv"])
def test_differentiating_through_the_iterations_matches_the_reference(tag):
    """BackwardMode.UNROLL / TRUNCATED on the generic path (nonlinear_least_squares.py:222-282: the Hessian is part of the graph):
    Gauss-Newton and adaptive (ellipsoidal) LM, with and without the convergence tests, against the REAL reference's gradients.
    (The fused pose-graph / bundle-adjustment paths: tests/test_unrolled_host.py.)"""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    from tests.simple_example_common import run_unrolled
    run_unrolled(th, load_golden("simple_example"), tag, "cpu", OracleKernels())
