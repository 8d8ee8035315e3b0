"""Dogleg (theseus/optimizer/nonlinear/dogleg.py + trust_region.py), the third NonlinearLeastSquares optimizer behind the
Linearization + LinearSolver boundary: theseus_amd's loop against trajectories recorded from the REAL reference
(tests/golden/pg_*_dogleg*.npz, oracle/gen_golden.py) -- on the CPU with the TEST stand-in kernels (host logic: the
device-side "all Gauss-Newton steps inside their regions" select, the block-wise ``Av``, rejection / replay), and on the GPU
through libtheseus_hip.so (-m gpu)."""
import ast
import warnings

import numpy as np
import pytest
import torch

from tests.helpers import load_golden

FP64 = ["pg_f64_dogleg", "pg_f64_dogleg_rejects"]


def _run(name, device, kernels=None, callback=False, **extra):
    import theseus_amd as th
    from tests.test_gpu_lm import build_objective
    g = load_golden(name)
    kw = ast.literal_eval(str(g["opt_kwargs"]))
    kw.pop("gauss_newton"), kw.pop("dogleg")
    obj, _ = build_objective(th, g, device=device)
    opt = th.Dogleg(obj, linear_solver_cls=th.HipCholeskySolver, max_iterations=kw.pop("max_iterations"),
                    step_size=kw.pop("step_size"), abs_err_tolerance=0.0, rel_err_tolerance=0.0,
                    linearization_kwargs=dict(kernels=kernels) if kernels is not None else None)
    taps = dict(delta=[], tr=[])
    okw = dict(track_err_history=True, **kw, **extra)
    if callback:   # any callback puts the loop on its synchronous path (one host decision per iteration)
        okw["end_iter_callback"] = lambda o, i, d, it: (taps["delta"].append(d.clone().cpu()),
                                                         taps["tr"].append(o._trust_region.view(-1).clone().cpu()))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=okw)
    final = torch.stack([sol[f"pose_{k}"] for k in range(int(g["P"]))], 1).cpu()
    return g, final, info, taps, opt


def _check(g, final, info, taps, tol):
    np.testing.assert_allclose(final.numpy(), g["final"], rtol=0, atol=tol)
    np.testing.assert_allclose(info.err_history.numpy(), g["err_history"], rtol=2e-5 if tol < 1e-6 else 2e-3)
    if taps["tr"]:
        np.testing.assert_array_equal(torch.stack(taps["tr"]).numpy(), g["trust_region"])   # the radii are powers of two
        for it, d in enumerate(taps["delta"]):
            np.testing.assert_allclose(d.numpy(), g["delta"][it], rtol=0, atol=tol * max(1.0, np.abs(g["delta"][it]).max()))


@pytest.mark.parametrize("name", FP64)
@pytest.mark.parametrize("callback", [False, True])
def test_dogleg_host_loop_matches_the_reference(name, callback):
    from tests.oracle_kernels import OracleKernels
    g, final, info, taps, opt = _run(name, "cpu", OracleKernels(), callback)
    _check(g, final, info, taps, 5e-8)
    assert info.iters_done == g["delta"].shape[0]
    if not callback:   # the sync-free path: the radius after the last iteration is the reference's
        np.testing.assert_array_equal(opt._trust_region.view(-1).numpy(), g["trust_region"][-1])


def test_block_wise_Av_equals_the_dense_product():
    """``Linearization.Av`` (dense_linearization.py:73-74) from the Jacobian blocks == A @ v with the materialised A."""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    from tests.test_gpu_lm import build_objective
    g = load_golden("pg_f64_dogleg")
    obj, _ = build_objective(th, g, device="cpu")
    lin = th.HipLinearization(obj, kernels=OracleKernels())
    lin.linearize()
    v = torch.randn(lin.H.shape[0], lin.num_cols, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    np.testing.assert_allclose(lin.Av(v).numpy(), (lin.A @ v.unsqueeze(2)).squeeze(2).numpy(), rtol=0, atol=1e-10)
    np.testing.assert_allclose(lin.A.numpy(), g["A0"], rtol=0, atol=1e-9 * np.abs(g["A0"]).max())


def test_trust_region_parameters_are_validated_like_the_reference():
    from tests.oracle_kernels import OracleKernels
    with pytest.raises(ValueError, match="Invalid parameters for TrustRegionMethod"):
        _run("pg_f64_dogleg", "cpu", OracleKernels(), shrink_threshold=0.9, expand_threshold=0.5)


def test_all_rejected_iteration_is_replayed_like_the_reference():
    """accept_threshold far above any gain ratio: every step of every problem is rejected, the reference retries twice without
    counting the iteration (nonlinear_optimizer.py:88, nonlinear_least_squares.py:186-194) -- same trajectory from the
    sync-free path (which replays) and from the synchronous one, and the oracle's."""
    from oracle import pose_graph as opg
    from tests.helpers import golden_problem
    from tests.oracle_kernels import OracleKernels
    kw = dict(accept_threshold=50.0, shrink_threshold=60.0, expand_threshold=70.0)
    g, fa, ia, _, oa = _run("pg_f64_dogleg", "cpu", OracleKernels(), False, **kw)
    _, fb, ib, _, ob = _run("pg_f64_dogleg", "cpu", OracleKernels(), True, **kw)
    assert torch.equal(fa, fb) and torch.equal(ia.err_history, ib.err_history) and ia.iters_done == ib.iters_done
    assert torch.equal(oa._trust_region, ob._trust_region)
    p, poses0, okw = golden_problem(g)
    okw.update(kw)
    fo, io = opg.lm_optimize(p, poses0, abs_err_tolerance=0.0, rel_err_tolerance=0.0, **okw)
    np.testing.assert_allclose(fa.numpy(), fo.numpy(), rtol=0, atol=1e-9)
    np.testing.assert_array_equal(fa.numpy(), g["poses0"])     # nothing was ever accepted
    assert ia.iters_done == io.iters_done


@pytest.mark.gpu
@pytest.mark.parametrize("name", FP64)
@pytest.mark.parametrize("callback", [False, True])
def test_dogleg_on_the_gpu_matches_the_reference(name, callback):
    g, final, info, taps, _ = _run(name, "cuda", None, callback)
    _check(g, final, info, taps, 1e-7)


@pytest.mark.gpu
def test_dogleg_fp32_on_the_gpu_inside_the_reference_band():
    """fp32: same criterion as the LM trajectories (tests/test_gpu_lm.py) -- at most 1.5x as far from the exact trajectory of
    the fp32 problem (fp64 oracle, fp32 Taylor thresholds) as the reference's own fp32 run."""
    from oracle import pose_graph as opg
    from tests.helpers import f32_thresholds, f32_truth_problem, golden_problem
    g, final, info, _, _ = _run("pg_f32_dogleg", "cuda")
    p, poses0, kw = golden_problem(g)
    p64, poses64 = f32_truth_problem(p, poses0)
    with f32_thresholds():
        exact, xinfo = opg.lm_optimize(p64, poses64, abs_err_tolerance=0.0, rel_err_tolerance=0.0, **kw)
    dev = (final.double() - exact).abs().max().item()
    dev_ref = (torch.from_numpy(g["final"]).double() - exact).abs().max().item()
    assert dev <= 1.5 * dev_ref + 1e-6 and dev <= 2e-4, (dev, dev_ref)
    hx = torch.stack(xinfo.err_history, 1)
    rel = ((info.err_history.double() - hx).abs() / hx).max().item()
    rel_ref = ((torch.from_numpy(g["err_history"]).double() - hx).abs() / hx).max().item()
    assert rel <= 1.5 * rel_ref + 1e-6, (rel, rel_ref)


@pytest.mark.parametrize("name", ["pg2_f64_lm", "pg3_f64_lm"])
def test_dogleg_on_se2_and_so3_graphs_matches_the_oracle(name):
    """The trust-region logic is group independent; the oracle's Dogleg (pinned to the reference on SE3 above) on the SE2 / SO3
    fixtures against theseus_amd.Dogleg with the stand-in kernels, incl. a convergence tolerance (the device-side flags)."""
    import theseus_amd as th
    from oracle import pose_graph as opg
    from tests.helpers import golden_problem
    from tests.oracle_kernels import OracleKernels
    from tests.test_gpu_lm import build_objective
    g = load_golden(name)
    p, poses0, _ = golden_problem(g)
    kw = dict(max_iterations=12, step_size=1.0, abs_err_tolerance=1e-9, rel_err_tolerance=1e-5)
    fo, io = opg.lm_optimize(p, poses0, dogleg=True, trust_region_init=0.3, **kw)
    obj, _ = build_objective(th, g, device="cpu")
    opt = th.Dogleg(obj, linear_solver_cls=th.HipCholeskySolver, linearization_kwargs=dict(kernels=OracleKernels()), **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(trust_region_init=0.3, track_err_history=True))
    final = torch.stack([sol[f"pose_{k}"] for k in range(int(g["P"]))], 1)
    np.testing.assert_allclose(final.numpy(), fo.numpy(), rtol=0, atol=1e-9)
    assert info.iters_done == io.iters_done and info.iters_done < 12        # stopped by the tolerance, at the same iteration
    np.testing.assert_array_equal(info.converged_iter.numpy(), io.converged_iter.numpy())
    hist = torch.stack(io.err_history, 1)
    np.testing.assert_allclose(info.err_history[:, :hist.shape[1]].numpy(), hist.numpy(), rtol=1e-9)


def test_dogleg_with_implicit_backward():
    """backward_mode="implicit" is optimizer independent (the last step is an undamped Gauss-Newton step,
    nonlinear_least_squares.py:121-135): Dogleg forward iterations + the fused implicit step, gradients against autograd through
    the oracle's implicit step from the same iterate."""
    import dataclasses
    import theseus_amd as th
    from oracle import pose_graph as opg
    from tests.helpers import golden_problem
    from tests.oracle_kernels import OracleKernels
    g = load_golden("pg_f64_implicit")
    p, poses0, kw = golden_problem(g)
    iters = 4
    t = torch.from_numpy
    meas = t(g["meas"]).clone().requires_grad_(True)
    obj = th.Objective(dtype=torch.float64)
    P = int(g["P"])
    poses = [th.SE3(tensor=poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
    for k in range(g["edges"].shape[0]):
        i, j = g["edges"][k].tolist()
        cw = th.DiagonalCostWeight(th.Variable(t(g["w_between"])[:, k].clone(), name=f"w_{k}"))
        obj.add(th.Between(poses[i], poses[j], th.SE3(tensor=meas[:, k], name=f"meas_{k}"), cw, name=f"between_{k}"))
    for k in range(g["prior_idx"].shape[0]):
        sw = th.ScaleCostWeight(th.Variable(t(g["w_prior"])[:, k, :1].clone(), name=f"pw_{k}"))
        obj.add(th.Difference(poses[int(g["prior_idx"][k])], th.SE3(tensor=t(g["prior_target"])[:, k].clone(), name=f"tgt_{k}"), sw,
                              name=f"prior_{k}"))
    opt = th.Dogleg(obj, linearization_kwargs=dict(kernels=OracleKernels()), max_iterations=iters, step_size=1.0,
                    abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(backward_mode="implicit", trust_region_init=1.0))
    final = torch.stack([sol[f"pose_{k}"] for k in range(P)], 1)
    coef = t(g["coef"])
    (coef * final).sum().backward()
    # the oracle: Dogleg for iters - 1 iterations under no_grad, then the implicit step with grad
    with torch.no_grad():
        x, _ = opg.lm_optimize(p, poses0, max_iterations=iters - 1, abs_err_tolerance=0.0, rel_err_tolerance=0.0, dogleg=True,
                               trust_region_init=1.0)
    m2 = p.meas.clone().requires_grad_(True)
    fo, _ = opg.implicit_final_step(dataclasses.replace(p, meas=m2), x)
    (coef * fo).sum().backward()
    np.testing.assert_allclose(final.detach().numpy(), fo.detach().numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(meas.grad.numpy(), m2.grad.numpy(), rtol=0, atol=1e-7 * float(m2.grad.abs().max()))
