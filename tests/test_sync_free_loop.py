"""The sync-free iteration path of the LM loop (theseus_amd/nonlinear.py: no adaptive damping / convergence test / callback
/ sharding -> the "a linear solve failed" flag stays on the device) takes the same steps as the path with one host
decision per iteration, and keeps the reference's failure semantics (nonlinear_least_squares.py:138-152: FAIL status,
variables untouched).  CPU, TEST stand-in kernels."""
import warnings

import numpy as np
import torch

from tests.helpers import load_golden


def _run(g, lazy, gauss_newton=False, tol=(0.0, 0.0), **okw):
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    from tests.test_gpu_lm import build_objective
    obj, poses = build_objective(th, g, device="cpu")
    cls = th.GaussNewton if gauss_newton else th.LevenbergMarquardt
    opt = cls(obj, linearization_kwargs=dict(kernels=OracleKernels()), abs_err_tolerance=tol[0], rel_err_tolerance=tol[1], **okw)
    calls = []
    kw = dict(track_err_history=True) if gauss_newton else dict(track_err_history=True, damping=1e-3)
    if not lazy:
        kw["end_iter_callback"] = lambda o, i, d, it: calls.append(it)   # any callback forces the per-iteration decision
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=kw)
    return torch.stack([sol[f"pose_{k}"] for k in range(int(g["P"]))], 1), info


def test_sync_free_iterations_equal_the_synchronised_loop():
    g = load_golden("pg_f64_lm")
    a, ia = _run(g, True, max_iterations=6, step_size=1.0)
    b, ib = _run(g, False, max_iterations=6, step_size=1.0)
    assert torch.equal(a, b) and torch.equal(ia.err_history, ib.err_history)
    np.testing.assert_allclose(a.numpy(), g["final"], rtol=0, atol=5e-8)
    assert ia.iters_done == ib.iters_done == 6


def test_sync_free_iterations_keep_the_failure_semantics():
    import theseus_amd as th
    g = dict(load_golden("pg_f64_gn"))
    g["w_between"] = g["w_between"] * 0.0
    g["w_prior"] = g["w_prior"] * 0.0          # H = 0: not positive definite (Gauss-Newton: no damping)
    for lazy in (True, False):
        a, info = _run(g, lazy, gauss_newton=True, max_iterations=4, step_size=1.0)
        assert all(s == th.NonlinearOptimizerStatus.FAIL for s in info.status), lazy
        np.testing.assert_array_equal(a.numpy(), g["poses0"])      # variables keep their values
        assert info.iters_done == 0 and torch.isinf(info.err_history[:, 1:]).all()


def test_convergence_is_counted_like_the_reference_on_both_paths():
    """Every problem converges at iteration k: the reference breaks out of its loop BEFORE counting that iteration
    (nonlinear_least_squares.py:202-203) -- err_history holds its error at [k + 1], converged_iter = k + 1, the iteration count
    stays k.  Same on the sync-free path (flags read once after the loop, lagged poll), the synchronous one and the oracle."""
    import theseus_amd as th
    from oracle import pose_graph as opg
    from tests.helpers import golden_problem
    g = load_golden("pg_f64_lm")
    tol = (1e-10, 1e-4)
    a, ia = _run(g, True, tol=tol, max_iterations=15, step_size=1.0)
    b, ib = _run(g, False, tol=tol, max_iterations=15, step_size=1.0)
    p, poses0, _ = golden_problem(g)
    fo, io = opg.lm_optimize(p, poses0, max_iterations=15, step_size=1.0, damping=1e-3, abs_err_tolerance=tol[0],
                             rel_err_tolerance=tol[1])
    assert 0 < io.iters_done < 14
    assert ia.iters_done == ib.iters_done == io.iters_done
    assert torch.equal(a, b) and torch.equal(ia.err_history, ib.err_history)
    assert torch.equal(ia.converged_iter, ib.converged_iter) and torch.equal(ia.converged_iter, io.converged_iter)
    k = io.iters_done
    assert torch.isfinite(ia.err_history[:, :k + 2]).all() and torch.isinf(ia.err_history[:, k + 2:]).all()
    np.testing.assert_allclose(a.numpy(), fo.numpy(), rtol=0, atol=1e-9)
    assert all(s == th.NonlinearOptimizerStatus.CONVERGED for s in ia.status) and list(ia.status) == list(ib.status)


def test_all_rejected_iteration_is_replayed_from_its_own_start():
    """an all-rejected step used to throw the whole sync-free pass away and replay optimize() from iteration 0 on the
    synchronous path.  Now the device keeps a snapshot of the start of the first all-rejected iteration (state, error, damping);
    only that iteration is replayed (the reference's retry branch) and the queueing resumes.  Problem 4 of the fixture rejects
    every step from iteration 6 on (rho = noise / noise once converged): as a batch of one, each of those is an ALL-rejection."""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    from tests.test_gpu_lm import build_objective
    g = dict(load_golden("pg_f64_lm_adaptive_rejects"))
    for k in ("poses0", "meas", "prior_target"):
        g[k] = g[k][4:5].copy()
    for k in ("w_between", "w_prior"):
        if g[k].shape[0] > 1:
            g[k] = g[k][4:5].copy()
    out = {}
    for lazy in (True, False):
        obj, _ = build_objective(th, g, device="cpu")
        K = OracleKernels()
        solves = []
        fac = K.chol_factor
        K.chol_factor = lambda *a, **k: (solves.append(1), fac(*a, **k))[1]
        opt = th.LevenbergMarquardt(obj, linearization_kwargs=dict(kernels=K), max_iterations=10, step_size=1.0,
                                    abs_err_tolerance=0.0, rel_err_tolerance=0.0)
        kw = dict(track_err_history=True, track_best_solution=True, damping=1e-4, adaptive_damping=True, damping_accept=0.9)
        if not lazy:
            kw["end_iter_callback"] = lambda o, i, d, it: None
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=kw)
        out[lazy] = (torch.stack([sol[f"pose_{k}"] for k in range(int(g["P"]))], 1), info, len(solves), opt._damping.clone())
    (a, ia, na, da), (b, ib, nb, db) = out[True], out[False]
    assert torch.equal(a, b) and torch.equal(ia.err_history, ib.err_history) and ia.iters_done == ib.iters_done == 10
    assert torch.equal(da, db) and torch.equal(ia.best_iter, ib.best_iter) and torch.equal(ia.best_err, ib.best_err)
    for k in ia.best_solution:
        assert torch.equal(ia.best_solution[k], ib.best_solution[k])
    assert nb > 10            # the synchronous run retried all-rejected steps (nonlinear_least_squares.py:358-359)
    # the sync-free run makes exactly the same attempts (the iteration counter lives on the device: nothing is replayed)
    assert na == nb, (na, nb)
