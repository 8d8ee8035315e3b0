"""The drop-in boundary against the REAL reference (this container only: needs /root/reference).

The reference's own ``th.LevenbergMarquardt`` loop drives ``theseus_amd.plugin.HipLinearization`` /
``HipCholeskySolver`` and must reproduce the trajectory it records with its own DenseLinearization +
CholeskyDenseSolver.  No GPU here, so the kernels behind the plugin are the TEST stand-in
(tests/oracle_kernels.py); the GPU tests check the HIP kernels against the same fixtures."""
import os
import sys

import numpy as np
import pytest
import torch

from tests.conftest import ROOT
from tests.helpers import golden_problem, load_golden

REF = "/root/reference"
pytestmark = [pytest.mark.reference, pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference")]


@pytest.fixture(scope="module")
def ref():
    for p in (os.path.join(ROOT, "oracle", "stubs"), REF, REF + "/torchlie", REF + "/torchkin"):
        if p not in sys.path:
            sys.path.append(p)  # appended: /root/reference has its own top-level ``tests`` package
    import warnings
    warnings.filterwarnings("ignore")
    import theseus as th
    import theseus_amd.plugin as thp
    return th, thp


def _objective(th, g):
    from oracle.gen_golden import build_reference_objective
    t = torch.from_numpy
    d = dict(P=int(g["P"]), edges=t(g["edges"]), meas=t(g["meas"]), w_between=t(g["w_between"]),
             prior_idx=t(g["prior_idx"]), prior_target=t(g["prior_target"]), w_prior=t(g["w_prior"]),
             poses=t(g["poses0"]))
    return build_reference_objective(th, d, d["poses"].dtype)


@pytest.mark.parametrize("name", ["pg_f64_lm", "pg_f64_lm_adaptive_ellips", "pg_f64_lm_adaptive_rejects", "pg_f64_gn"])
def test_reference_loop_drives_the_plugin(ref, name):
    th, thp = ref
    from tests.oracle_kernels import OracleKernels
    g = load_golden(name)
    _, _, kw = golden_problem(g)
    obj, poses = _objective(th, g)
    gn = kw.pop("gauss_newton")
    okw = dict(max_iterations=kw.pop("max_iterations"), step_size=kw.pop("step_size"))
    cls = th.GaussNewton if gn else th.LevenbergMarquardt
    opt = cls(obj, linear_solver_cls=thp.HipCholeskySolver, linearization_cls=thp.HipLinearization,
              linearization_kwargs=dict(kernels=OracleKernels()), vectorize=True,
              abs_err_tolerance=0.0, rel_err_tolerance=0.0, **okw)
    lin = opt.linear_solver.linearization
    assert lin.var_start_cols == list(g["var_start_cols"]) and lin.num_rows == int(g["num_rows"])
    with torch.no_grad():
        info = opt.optimize(track_err_history=True, **kw)
    final = torch.stack([p.tensor for p in poses], 1).numpy()
    # converged problems' late accept/reject decisions are coin flips (tests/test_gpu_lm.py:well_conditioned_steps)
    from tests.test_gpu_lm import well_conditioned_steps
    ok = well_conditioned_steps(g, g["delta"].shape[0])
    slack = 2.0 * (np.abs(g["delta"]).max(axis=2) * ~ok).sum(axis=0)
    assert (np.abs(final - g["final"]).reshape(final.shape[0], -1).max(1) <= 5e-8 + slack).all()
    assert all(s == th.NonlinearOptimizerStatus.MAX_ITERATIONS for s in info.status)
    # the properties the reference reads from a linearization
    lin.linearize()
    ref_lin = th.optimizer.DenseLinearization(obj)
    ref_lin.linearize()
    np.testing.assert_allclose(lin.AtA.numpy(), ref_lin.AtA.numpy(), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(lin.Atb.numpy(), ref_lin.Atb.numpy(), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(lin.A.numpy(), ref_lin.A.numpy(), rtol=1e-10, atol=1e-10)
    v = torch.randn(lin.AtA.shape[0], lin.num_cols, dtype=torch.float64)
    np.testing.assert_allclose(lin.diagonal_scaling(v).numpy(), ref_lin.diagonal_scaling(v).numpy(), rtol=1e-10)
    np.testing.assert_allclose(lin.Av(v).numpy(), ref_lin.Av(v).numpy(), rtol=1e-9, atol=1e-9)


def test_plugin_through_theseus_layer_and_failure_path(ref):
    th, thp = ref
    from tests.oracle_kernels import OracleKernels
    g = load_golden("pg_f64_lm")
    obj, poses = _objective(th, g)
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=thp.HipCholeskySolver,
                                linearization_kwargs=dict(kernels=OracleKernels()), max_iterations=6,
                                abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    layer = th.TheseusLayer(opt)
    with torch.no_grad():
        sol, info = layer.forward(optimizer_kwargs=dict(damping=1e-3))
    final = torch.stack([sol[f"pose_{k}"] for k in range(int(g["P"]))], 1).numpy()
    np.testing.assert_allclose(final, g["final"], rtol=0, atol=5e-8)
    # non positive definite system -> RuntimeError from solve() -> the reference loop reports FAIL
    for cf in obj.cost_functions.values():
        for v in cf.weight.aux_vars:
            v.update(torch.zeros_like(v.tensor))
    with torch.no_grad(), pytest.warns(RuntimeWarning):
        info = th.GaussNewton(obj, linear_solver_cls=thp.HipCholeskySolver,
                              linearization_kwargs=dict(kernels=OracleKernels()), max_iterations=3).optimize()
    assert all(s == th.NonlinearOptimizerStatus.FAIL for s in info.status)


def test_unsupported_cost_function_is_refused(ref):
    th, thp = ref
    obj = th.Objective(dtype=torch.float64)
    a, b = th.Vector(2, name="a", dtype=torch.float64), th.Vector(2, name="b", dtype=torch.float64)
    obj.add(th.Difference(a, b, th.ScaleCostWeight(torch.tensor(1.0, dtype=torch.float64)), name="d"))
    with pytest.raises(NotImplementedError):
        th.LevenbergMarquardt(obj, linear_solver_cls=thp.HipCholeskySolver)
