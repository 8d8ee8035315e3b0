"""Convergence tests switched ON (nonlinear_optimizer.py:110-119, nonlinear_least_squares.py:196-203), against runs of the REAL
reference (tests/golden/pg_f64_lm_converges.npz: problems converge at different iterations, converged ones are frozen, the loop
stops when all have; pg_f64_lm_partly_converges.npz: the iteration budget ends first): final poses, err_history with its inf
tail, status per problem, converged_iter -- the oracle, theseus_amd's loop on both of its paths (CPU, stand-in kernels) and on
the GPU."""
import warnings

import numpy as np
import pytest
import torch

from tests.helpers import golden_problem, load_golden

CASES = ["pg_f64_lm_converges", "pg_f64_lm_partly_converges"]


def _expect(g):
    return g["final"], g["err_history"], g["status"].tolist(), g["converged_iter"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_the_reference(name):
    from oracle import pose_graph as opg
    g = load_golden(name)
    p, poses0, kw = golden_problem(g)
    final, info = opg.lm_optimize(p, poses0, **kw)
    ref_final, ref_hist, _, ref_conv = _expect(g)
    np.testing.assert_allclose(final.numpy(), ref_final, rtol=0, atol=5e-8)
    hist = torch.stack(info.err_history, 1).numpy()
    k = hist.shape[1]
    assert np.isfinite(ref_hist[:, :k]).all() and np.isinf(ref_hist[:, k:]).all()
    np.testing.assert_allclose(hist, ref_hist[:, :k], rtol=2e-5)
    np.testing.assert_array_equal(info.converged_iter.numpy(), ref_conv)


def _run(name, device, kernels, callback):
    import theseus_amd as th
    from tests.test_gpu_lm import build_objective
    g = load_golden(name)
    _, _, kw = golden_problem(g)
    kw.pop("gauss_newton")
    okw = {k: kw.pop(k) for k in ("max_iterations", "step_size", "abs_err_tolerance", "rel_err_tolerance")}
    obj, _ = build_objective(th, g, device=device)
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipCholeskySolver,
                                linearization_kwargs=dict(kernels=kernels) if kernels is not None else None, **okw)
    calls = []
    extra = dict(end_iter_callback=lambda o, i, d, it: calls.append(it)) if callback else {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(track_err_history=True, **kw, **extra))
    final = torch.stack([sol[f"pose_{k}"] for k in range(int(g["P"]))], 1).cpu()
    return g, final, info, calls


def _check(g, final, info, calls, callback):
    ref_final, ref_hist, ref_status, ref_conv = _expect(g)
    np.testing.assert_allclose(final.numpy(), ref_final, rtol=0, atol=1e-7)
    hist = info.err_history.numpy()
    assert hist.shape == ref_hist.shape
    np.testing.assert_array_equal(np.isfinite(hist), np.isfinite(ref_hist))          # the same inf tail
    np.testing.assert_allclose(hist[np.isfinite(ref_hist)], ref_hist[np.isfinite(ref_hist)], rtol=2e-5)
    assert [int(s.value) for s in info.status] == ref_status
    np.testing.assert_array_equal(info.converged_iter.numpy(), ref_conv)
    if callback:   # the reference calls end_iter_callback after every counted iteration, not after the one it breaks out of
        assert calls == list(range(g["delta"].shape[0]))
    assert info.iters_done == g["delta"].shape[0]


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("callback", [False, True])
def test_host_loop_matches_the_reference(name, callback):
    from tests.oracle_kernels import OracleKernels
    g, final, info, calls = _run(name, "cpu", OracleKernels(), callback)
    _check(g, final, info, calls, callback)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("callback", [False, True])
def test_gpu_loop_matches_the_reference(name, callback):
    g, final, info, calls = _run(name, "cuda", None, callback)
    _check(g, final, info, calls, callback)


def _run_best(device, kernels):
    import theseus_amd as th
    from tests.test_gpu_lm import build_objective
    g = load_golden("pg_f64_gn_best")
    obj, _ = build_objective(th, g, device=device)
    opt = th.GaussNewton(obj, linear_solver_cls=th.HipCholeskySolver, max_iterations=1, step_size=1.0, abs_err_tolerance=0.0,
                         rel_err_tolerance=0.0, linearization_kwargs=dict(kernels=kernels) if kernels is not None else None)
    sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(track_best_solution=True, track_err_history=True))
    P = int(g["P"])
    final = torch.stack([sol[f"pose_{k}"] for k in range(P)], 1).cpu()
    best = torch.stack([info.best_solution[f"pose_{k}"] for k in range(P)], 1).cpu()
    return g, final, best, info


def _check_best(g, final, best, info):
    """track_best_solution (nonlinear_optimizer.py:184-213): one Gauss-Newton step from far away leaves two of the six problems
    ABOVE their starting error -- their best solution is the start, the others' is the new iterate."""
    np.testing.assert_allclose(final.numpy(), g["final"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(best.numpy(), g["best_solution"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(info.best_err.cpu().numpy(), g["best_err"], rtol=2e-5)
    np.testing.assert_array_equal(info.best_iter.numpy(), g["best_iter"])
    worse = g["err_history"][:, 1] > g["err_history"][:, 0]
    assert worse.sum() == 2
    np.testing.assert_array_equal(best.numpy()[worse], g["poses0"][worse])      # bit for bit the starting values


def test_best_solution_matches_the_reference():
    from tests.oracle_kernels import OracleKernels
    _check_best(*_run_best("cpu", OracleKernels()))


@pytest.mark.gpu
def test_best_solution_on_the_gpu_matches_the_reference():
    _check_best(*_run_best("cuda", None))
