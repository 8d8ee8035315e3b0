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
