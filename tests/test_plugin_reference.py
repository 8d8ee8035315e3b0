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

# THX_REFERENCE_ROOT / THX_PLUGIN_DEVICE=cuda: the same tests with the REAL HIP kernels behind the plugin, on a GPU box that
# was handed a scratch copy of the reference for the occasion (tools/dropin_gpu.sh; the copy is never committed).
REF = os.environ.get("THX_REFERENCE_ROOT", "/root/reference")
DEVICE = os.environ.get("THX_PLUGIN_DEVICE", "cpu")
pytestmark = [pytest.mark.reference, pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference")]
cpu_only = pytest.mark.skipif(DEVICE != "cpu", reason="builds its tensors on the CPU / spies on the stand-in")


def _kernels():
    """linearization_kwargs: the TEST stand-in on the CPU, libtheseus_hip.so (the default) on a GPU."""
    if DEVICE == "cpu":
        from tests.oracle_kernels import OracleKernels
        return dict(kernels=OracleKernels())
    return {}


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
             poses=t(g["poses0"]), group=str(g["group"]) if "group" in g else "SE3")
    obj, poses = build_reference_objective(th, d, d["poses"].dtype)
    if DEVICE != "cpu":
        obj.to(DEVICE)
    return obj, poses


@pytest.mark.parametrize("name", ["pg_f64_lm", "pg_f64_lm_adaptive_ellips", "pg_f64_lm_adaptive_rejects", "pg_f64_gn",
                                  "pg2_f64_lm", "pg2_f64_lm_adaptive", "pg3_f64_lm", "pg3_f64_lm_adaptive",
                                  "pgso2_f64_lm", "pgso2_f64_lm_adaptive",                              # th.SO2 variables: fused too
                                  "pg_f64_dogleg", "pg_f64_dogleg_rejects"])   # th.Dogleg: Av() from the Jacobian blocks
def test_reference_loop_drives_the_plugin(ref, name):
    th, thp = ref
    from tests.oracle_kernels import OracleKernels
    g = load_golden(name)
    _, _, kw = golden_problem(g)
    obj, poses = _objective(th, g)
    gn = kw.pop("gauss_newton", False)
    okw = dict(max_iterations=kw.pop("max_iterations"), step_size=kw.pop("step_size"))
    cls = th.Dogleg if kw.pop("dogleg", False) else (th.GaussNewton if gn else th.LevenbergMarquardt)
    opt = cls(obj, linear_solver_cls=thp.HipCholeskySolver, linearization_cls=thp.HipLinearization,
              linearization_kwargs=_kernels(), vectorize=True,
              abs_err_tolerance=0.0, rel_err_tolerance=0.0, **okw)
    lin = opt.linear_solver.linearization
    assert lin.var_start_cols == list(g["var_start_cols"]) and lin.num_rows == int(g["num_rows"])
    with torch.no_grad():
        info = opt.optimize(track_err_history=True, **kw)
    final = torch.stack([p.tensor for p in poses], 1).detach().cpu().numpy()
    # converged problems' late accept/reject decisions are coin flips (tests/test_gpu_lm.py:well_conditioned_steps)
    from tests.test_gpu_lm import well_conditioned_steps
    ok = well_conditioned_steps(g, g["delta"].shape[0])
    slack = 2.0 * (np.abs(g["delta"]).max(axis=2) * ~ok).sum(axis=0)
    assert (np.abs(final - g["final"]).reshape(final.shape[0], -1).max(1) <= 5e-8 + slack).all()
    assert all(s == th.NonlinearOptimizerStatus.MAX_ITERATIONS for s in info.status)
    if "trust_region" in g:
        np.testing.assert_array_equal(opt._trust_region.view(-1).cpu().numpy(), g["trust_region"][-1])
    # the properties the reference reads from a linearization
    lin.linearize()
    ref_lin = th.optimizer.DenseLinearization(obj)
    ref_lin.linearize()
    np.testing.assert_allclose(lin.AtA.detach().cpu().numpy(), ref_lin.AtA.detach().cpu().numpy(), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(lin.Atb.detach().cpu().numpy(), ref_lin.Atb.detach().cpu().numpy(), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(lin.A.detach().cpu().numpy(), ref_lin.A.detach().cpu().numpy(), rtol=1e-10, atol=1e-10)
    v = torch.randn(lin.AtA.shape[0], lin.num_cols, dtype=torch.float64).to(DEVICE)
    np.testing.assert_allclose(lin.diagonal_scaling(v).detach().cpu().numpy(), ref_lin.diagonal_scaling(v).detach().cpu().numpy(), rtol=1e-10)
    np.testing.assert_allclose(lin.Av(v).detach().cpu().numpy(), ref_lin.Av(v).detach().cpu().numpy(), rtol=1e-9, atol=1e-9)


def test_plugin_through_theseus_layer_and_failure_path(ref):
    th, thp = ref
    from tests.oracle_kernels import OracleKernels
    g = load_golden("pg_f64_lm")
    obj, poses = _objective(th, g)
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=thp.HipCholeskySolver,
                                linearization_kwargs=_kernels(), max_iterations=6,
                                abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    layer = th.TheseusLayer(opt)
    with torch.no_grad():
        sol, info = layer.forward(optimizer_kwargs=dict(damping=1e-3))
    final = torch.stack([sol[f"pose_{k}"] for k in range(int(g["P"]))], 1).detach().cpu().numpy()
    np.testing.assert_allclose(final, g["final"], rtol=0, atol=5e-8)
    # non positive definite system -> RuntimeError from solve() -> the reference loop reports FAIL
    for cf in obj.cost_functions.values():
        for v in cf.weight.aux_vars:
            v.update(torch.zeros_like(v.tensor))
    with torch.no_grad(), pytest.warns(RuntimeWarning):
        info = th.GaussNewton(obj, linear_solver_cls=thp.HipCholeskySolver,
                              linearization_kwargs=_kernels(), max_iterations=3).optimize()
    assert all(s == th.NonlinearOptimizerStatus.FAIL for s in info.status)


def test_lagged_failure_check_reports_fail_one_solve_late_and_changes_nothing_otherwise(ref):
    """``linear_solver_kwargs=dict(lagged_failure_check=True)``: no host look at the factorisation's info inside solve().  Same
    trajectory on a healthy problem; a failed factorisation drops its step on the device and is raised by the NEXT solve() --
    the reference loop still ends in FAIL (nonlinear_least_squares.py:138-152) with the variables where they were."""
    th, thp = ref
    g = load_golden("pg_f64_lm")
    obj, poses = _objective(th, g)
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=thp.HipCholeskySolver, linearization_kwargs=_kernels(),
                                linear_solver_kwargs=dict(lagged_failure_check=True), max_iterations=6,
                                abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    assert opt.linear_solver._lagged
    with torch.no_grad():
        sol, info = th.TheseusLayer(opt).forward(optimizer_kwargs=dict(damping=1e-3))
    final = torch.stack([sol[f"pose_{k}"] for k in range(int(g["P"]))], 1).detach().cpu().numpy()
    np.testing.assert_allclose(final, g["final"], rtol=0, atol=5e-8)
    for cf in obj.cost_functions.values():
        for v in cf.weight.aux_vars:
            v.update(torch.zeros_like(v.tensor))
    before = [p.tensor.clone() for p in poses]
    with torch.no_grad(), pytest.warns(RuntimeWarning):
        info = th.GaussNewton(obj, linear_solver_cls=thp.HipCholeskySolver, linearization_kwargs=_kernels(),
                              linear_solver_kwargs=dict(lagged_failure_check=True), max_iterations=3,
                              abs_err_tolerance=0.0, rel_err_tolerance=0.0).optimize()
    assert all(s == th.NonlinearOptimizerStatus.FAIL for s in info.status)
    for p, b in zip(poses, before):
        assert torch.equal(p.tensor, b)            # the failed steps were dropped, not applied


@cpu_only
@pytest.mark.parametrize("optimizer", ["LevenbergMarquardt", "Dogleg"])
def test_non_pose_graph_objective_takes_the_generic_path(ref, optimizer):
    """Vector variables + Difference: not an SE3 pose graph -> generic block assembly, same result as the reference
    (th.Dogleg also reads ``Av`` of the generic path: block-wise, no dense Jacobian)."""
    th, thp = ref
    from tests.oracle_kernels import OracleKernels
    out = {}
    for tag in ("ref", "ours"):
        obj = th.Objective(dtype=torch.float64)
        a = th.Vector(tensor=torch.tensor([[1.0, -2.0], [0.5, 3.0]], dtype=torch.float64), name="a")
        b = th.Vector(tensor=torch.tensor([[0.0, 1.0], [2.0, 2.0]], dtype=torch.float64), name="b")
        tgt = th.Vector(tensor=torch.tensor([[4.0, 4.0]], dtype=torch.float64), name="t")
        w = th.ScaleCostWeight(torch.tensor(2.0, dtype=torch.float64))
        obj.add(th.Difference(a, tgt, w, name="d0"))
        obj.add(th.Between(a, b, th.Vector(tensor=torch.tensor([[1.0, 1.0]], dtype=torch.float64), name="m"), w, name="d1"))
        kw = {} if tag == "ref" else dict(linear_solver_cls=thp.HipCholeskySolver,
                                          linearization_kwargs=_kernels())
        opt = getattr(th, optimizer)(obj, max_iterations=5, **kw)
        obj.update()
        with torch.no_grad():
            opt.optimize(**(dict(damping=0.1) if optimizer == "LevenbergMarquardt" else dict(trust_region_init=1.0)))
        out[tag] = torch.cat([a.tensor, b.tensor], 1)
        if tag == "ours":
            assert not opt.linear_solver.linearization.fused
    np.testing.assert_allclose(out["ours"].detach().cpu().numpy(), out["ref"].detach().cpu().numpy(), rtol=1e-12, atol=1e-12)


def _quadratic_fit(th, B=16, N=20, seed=0):
    """examples/simple_example.py (BASELINE.json configs[0]): fit y = v exp(x), AutoDiffCostFunction on a Vector."""
    gen = torch.Generator().manual_seed(seed)
    x_true = torch.linspace(-1, 1, N, dtype=torch.float64).view(1, -1).repeat(B, 1)
    c_true = 0.5 + 0.2 * torch.rand(B, 1, dtype=torch.float64, generator=gen)
    y = c_true * torch.exp(x_true)
    x = th.Variable((x_true + 0.05 * torch.randn(B, N, dtype=torch.float64, generator=gen)), name="x")
    yv = th.Variable(y, name="y")
    v = th.Vector(1, name="v", dtype=torch.float64)

    def error_fn(optim_vars, aux_vars):
        xx, yy = aux_vars
        return yy.tensor - optim_vars[0].tensor * torch.exp(xx.tensor)

    obj = th.Objective(dtype=torch.float64)
    obj.add(th.AutoDiffCostFunction([v], error_fn, N, aux_vars=[x, yv],
                                    cost_weight=th.ScaleCostWeight(torch.tensor(1.0, dtype=torch.float64))))
    return obj, x.tensor.clone()


@cpu_only
def test_generic_path_runs_simple_example_config0(ref):
    """BASELINE.json configs[0]: quadratic fit, batch 16, GaussNewton + dense Cholesky, implicit backward through
    TheseusLayer -- the reference's own loop and AutoDiffCostFunction, our Linearization / LinearSolver behind it
    (generic block assembly; stand-in kernels here, HIP kernels on a GPU)."""
    th, thp = ref
    from tests.oracle_kernels import OracleKernels
    res = {}
    for tag in ("ref", "ours"):
        obj, x0 = _quadratic_fit(th)
        kw = {} if tag == "ref" else dict(linear_solver_cls=thp.HipCholeskySolver,
                                          linearization_kwargs=_kernels())
        layer = th.TheseusLayer(th.GaussNewton(obj, max_iterations=10, **kw))
        phi = x0.clone().requires_grad_(True)
        sol, info = layer.forward(input_tensors={"x": phi, "v": torch.ones(16, 1, dtype=torch.float64)},
                                  optimizer_kwargs={"backward_mode": "implicit"})
        loss = ((sol["v"] - 0.5) ** 2).mean()
        loss.backward()
        res[tag] = (sol["v"].detach(), loss.item(), phi.grad.clone(), info)
        if tag == "ours":
            assert not layer.optimizer.linear_solver.linearization.fused
    np.testing.assert_allclose(res["ours"][0].detach().cpu().numpy(), res["ref"][0].detach().cpu().numpy(), rtol=1e-10, atol=1e-12)
    assert abs(res["ours"][1] - res["ref"][1]) < 1e-12
    np.testing.assert_allclose(res["ours"][2].detach().cpu().numpy(), res["ref"][2].detach().cpu().numpy(), rtol=1e-8, atol=1e-12)


@cpu_only
def test_fused_path_is_differentiable_through_the_reference_loop(ref):
    """backward_mode="implicit" of the REAL loop with the plugin on an SE3 pose graph: Atb's backward is the fused
    VJP, the solve's backward the cached-factor solve; gradients equal the ones the reference recorded."""
    th, thp = ref
    from tests.oracle_kernels import OracleKernels
    g = load_golden("pg_f64_implicit")
    t = torch.from_numpy
    _, _, kw = golden_problem(g)
    kw.pop("gauss_newton")
    P = int(g["P"])
    leaves = dict(meas=t(g["meas"]).requires_grad_(True), w_between=t(g["w_between"]).requires_grad_(True),
                  prior_target=t(g["prior_target"]).requires_grad_(True),
                  w_prior=t(g["w_prior"])[:, :, :1].clone().requires_grad_(True))
    obj = th.Objective(dtype=torch.float64)
    poses = [th.SE3(tensor=t(g["poses0"])[:, k].clone(), name=f"pose_{k}") for k in range(P)]
    for k in range(g["edges"].shape[0]):
        i, j = g["edges"][k].tolist()
        obj.add(th.Between(poses[i], poses[j], th.SE3(tensor=leaves["meas"][:, k], name=f"meas_{k}"),
                           th.DiagonalCostWeight(th.Variable(leaves["w_between"][:, k], name=f"w_{k}")), name=f"between_{k}"))
    for k in range(g["prior_idx"].shape[0]):
        obj.add(th.Difference(poses[int(g["prior_idx"][k])], th.SE3(tensor=leaves["prior_target"][:, k], name=f"tgt_{k}"),
                              th.ScaleCostWeight(th.Variable(leaves["w_prior"][:, k], name=f"pw_{k}")), name=f"prior_{k}"))
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=thp.HipCholeskySolver,
                                linearization_kwargs=_kernels(), max_iterations=kw.pop("max_iterations"),
                                step_size=kw.pop("step_size"), abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    assert opt.linear_solver.linearization.fused
    sol, _ = th.TheseusLayer(opt).forward(optimizer_kwargs=dict(backward_mode="implicit", **kw))
    final = torch.stack([sol[f"pose_{k}"] for k in range(P)], 1)
    (t(g["coef"]) * final).sum().backward()
    np.testing.assert_allclose(final.detach().detach().cpu().numpy(), g["final"], rtol=0, atol=1e-7)
    for key, refk in (("meas", "grad_meas"), ("w_between", "grad_w_between"), ("prior_target", "grad_prior_target"),
                      ("w_prior", "grad_w_prior")):
        want = g[refk]
        np.testing.assert_allclose(leaves[key].grad.detach().cpu().numpy(), want, rtol=0, atol=2e-6 * np.abs(want).max(), err_msg=key)


@cpu_only
def test_unmodified_reference_example_runs_on_the_plugin(ref, monkeypatch):
    """examples/pose_graph/pose_graph_synthetic.py -- UNMODIFIED, imported from /root/reference -- with
    ``inner_optim.linear_solver_cls = HipCholeskySolver`` (the one name a user changes): Welsch RobustCostFunction,
    adaptive LM and implicit backward go through the plugin (fused assembly, cached-factor backward solve, fused VJP
    incl. the log_loss_radius gradient) and reproduce the reference's published known-answer test
    (tests/theseus_tests/test_pgo_benchmark.py:34-39) at its own tolerance."""
    th, thp = ref
    import logging
    import theseus_amd.kernels as tk
    from omegaconf import OmegaConf
    from tests.oracle_kernels import OracleKernels
    import examples.pose_graph.pose_graph_synthetic as pgo
    from oracle.gen_golden import PGO_KAT_LOSSES
    standin = OracleKernels()
    calls = {"pg_assemble": 0, "pg_vjp": 0}
    for name in calls:
        def spy(*a, _f=getattr(standin, name), _n=name, **k):
            calls[_n] += 1
            return _f(*a, **k)
        setattr(standin, name, spy)
    monkeypatch.setattr(tk, "_default", standin)                  # no GPU here: the TEST stand-in
    monkeypatch.setattr(th, "HipCholeskySolver", thp.HipCholeskySolver, raising=False)
    monkeypatch.chdir("/tmp")
    logging.disable(logging.CRITICAL)
    try:
        cfg = OmegaConf.load(REF + "/examples/configs/pose_graph/pose_graph_synthetic.yaml")
        cfg.outer_optim.num_epochs = 1
        cfg.outer_optim.max_num_batches = 4
        cfg.batch_size = 16
        cfg.num_poses = 64
        cfg.profile = False
        cfg.savemat = False
        cfg.inner_optim.optimizer_kwargs.verbose = False
        cfg.inner_optim.reg_w = float(cfg.inner_optim.reg_w)
        cfg.inner_optim.linear_solver_cls = "HipCholeskySolver"
        cfg.device = "cpu"
        losses = pgo.run(cfg)[0]
    finally:
        logging.disable(logging.NOTSET)
    for got, want in zip(losses, PGO_KAT_LOSSES):
        assert got == pytest.approx(want, rel=1e-10, abs=1e-10), (losses, PGO_KAT_LOSSES)
    assert calls["pg_assemble"] >= 40 and calls["pg_vjp"] == 4, calls   # the FUSED path ran, once per outer backward


@cpu_only
def test_unmodified_reference_bundle_adjustment_example_runs_on_the_plugin(ref, monkeypatch, tmp_path):
    """examples/bundle_adjustment.py -- UNMODIFIED, imported from /root/reference, its own config -- learns ``log_loss_radius``
    through ``backward_mode="implicit"`` (Adam on the outer loss, :184-215).  The example builds its optimizer as
    ``getattr(th, cfg.inner_optim.optimizer_cls)(objective, max_iterations=..., step_size=...)``; the one thing a user adds is
    ``linear_solver_cls`` -- injected here through the name the config resolves.  Same outer losses and the same learned radius,
    epoch by epoch, as the example on the reference's own dense path."""
    th, thp = ref
    import functools
    import logging
    import random
    import theseus_amd.kernels as tk
    from omegaconf import OmegaConf
    from tests.oracle_kernels import OracleKernels
    import examples.bundle_adjustment as ba_ex
    cfg = OmegaConf.load(REF + "/examples/configs/bundle_adjustment.yaml")
    cfg.outer_optim.num_epochs = 2
    cfg.inner_optim.verbose = False
    cfg.inner_optim.max_iters = 3
    # 60 of the config's 200 points: the REFERENCE's dense run keeps one copy of A per slice-assignment alive for its backward
    # (dense_linearization.py:44-55 under autograd) -- 58 GB of host memory at the config's ~1600 observations, 3 GB at ~480
    cfg.num_points = 60
    cfg.inner_optim.reg_w = float(cfg.inner_optim.reg_w)
    standin = OracleKernels()
    calls = {"ba_assemble": 0, "ba_vjp": 0}
    for name in calls:
        def spy(*a, _f=getattr(standin, name), _n=name, **k):
            calls[_n] += 1
            return _f(*a, **k)
        setattr(standin, name, spy)

    def run(tag, plugin):
        torch.manual_seed(cfg.seed), np.random.seed(cfg.seed), random.seed(cfg.seed)   # what the example's main() does
        out = tmp_path / tag
        out.mkdir()
        with monkeypatch.context() as m:
            if plugin:
                m.setattr(tk, "_default", standin)                  # no GPU here: the TEST stand-in
                m.setattr(th, "GaussNewton", functools.partial(th.GaussNewton, linear_solver_cls=thp.HipSchurSolver))
            ba_ex.run(cfg, out)
        res = [torch.load(out / f"results_epoch{e}.pt", weights_only=False) for e in range(cfg.outer_optim.num_epochs)]
        return [r["loss"] for r in res], [float(r["log_loss_radius"]) for r in res]
    logging.disable(logging.CRITICAL)
    try:
        ref_losses, ref_radius = run("reference", False)
        losses, radius = run("plugin", True)
    finally:
        logging.disable(logging.NOTSET)
    assert calls["ba_assemble"] >= 2 * 3 and calls["ba_vjp"] == 2, calls      # the fused BA path ran, one VJP per outer backward
    assert len(set(ref_radius)) == 2                                          # the radius is being learned
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(radius, ref_radius, rtol=1e-9)


@pytest.mark.parametrize("name", ["ba_f64_lm", "ba_f64_gn", "ba_f64_camcam_lm"])
def test_reference_loop_drives_the_bundle_adjustment_plugin(ref, name):
    """The REAL th.LevenbergMarquardt / th.GaussNewton loop with theseus_amd.plugin.HipSchurSolver on a bundle-adjustment
    objective built from the reference's own classes (th.eb.Reprojection, th.RobustCostFunction, th.Difference on SE3 and
    Point3) reproduces the trajectory the reference recorded with DenseLinearization + CholeskyDenseSolver."""
    th, thp = ref
    import ast
    from tests.ba_common import build_ba_objective
    from tests.oracle_kernels import OracleKernels

    class RefNames:  # build_ba_objective speaks theseus_amd's names: map them onto the reference's
        Objective, SE3, Point3, Point2, Vector, Difference = th.Objective, th.SE3, th.Point3, th.Point2, th.Vector, th.Difference
        ScaleCostWeight, RobustCostFunction, HuberLoss, WelschLoss = th.ScaleCostWeight, th.RobustCostFunction, th.HuberLoss, th.WelschLoss
        Reprojection, Between, DiagonalCostWeight, Variable = th.eb.Reprojection, th.Between, th.DiagonalCostWeight, th.Variable
    g = load_golden(name)
    obj, cam_v, pt_v = build_ba_objective(RefNames, g, DEVICE)
    if DEVICE != "cpu":
        obj.to(DEVICE)
    obj.update()   # resolves the batch size (TheseusLayer.forward does this for the user)
    kw = ast.literal_eval(str(g["opt_kwargs"]))
    gn = kw.pop("gauss_newton")
    opt = (th.GaussNewton if gn else th.LevenbergMarquardt)(
        obj, linear_solver_cls=thp.HipSchurSolver, linearization_kwargs=_kernels(), vectorize=True,
        abs_err_tolerance=0.0, rel_err_tolerance=0.0, max_iterations=kw.pop("max_iterations"), step_size=kw.pop("step_size"))
    with torch.no_grad():
        info = opt.optimize(track_err_history=True, **kw)
    cams = torch.stack([v.tensor for v in cam_v], 1).detach().cpu().numpy()
    used = sorted(set(g["obs_pt"].tolist()))
    pts = torch.stack([pt_v[i].tensor for i in used], 1).detach().cpu().numpy()
    np.testing.assert_allclose(cams, g["final_cams"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(pts, g["final_pts"][:, used], rtol=0, atol=1e-5)
    k = min(info.err_history.shape[1], g["err_history"].shape[1])
    np.testing.assert_allclose(info.err_history[:, :k].detach().cpu().numpy(), g["err_history"][:, :k], rtol=1e-6)


def test_bundle_adjustment_implicit_backward_through_the_reference_loop(ref):
    """examples/bundle_adjustment.py:184-215 learns log_loss_radius through backward_mode="implicit": the REAL TheseusLayer /
    LevenbergMarquardt with theseus_amd.plugin.HipSchurSolver -- forward iterations under no_grad, the grad-enabled last
    Gauss-Newton step with ``Atb`` differentiable (thx_ba_vjp) and the solve's backward on the cached Schur factor, the
    reference's own retraction in between -- reproduces the gradients the reference recorded with its dense path, all nine
    groups (features, calibration, radius, weights, prior targets)."""
    th, thp = ref
    from tests.ba_common import run_ba_implicit

    class RefNames:
        Objective, SE3, Point3, Point2, Vector, Variable, Difference = (th.Objective, th.SE3, th.Point3, th.Point2, th.Vector,
                                                                       th.Variable, th.Difference)
        ScaleCostWeight, RobustCostFunction, HuberLoss, WelschLoss = th.ScaleCostWeight, th.RobustCostFunction, th.HuberLoss, th.WelschLoss
        Reprojection = th.eb.Reprojection
        LevenbergMarquardt, TheseusLayer = th.LevenbergMarquardt, th.TheseusLayer
    g = load_golden("ba_f64_implicit")
    out = run_ba_implicit(RefNames, g, device=DEVICE,
                          opt_kwargs=dict(linear_solver_cls=thp.HipSchurSolver, linearization_kwargs=_kernels(), vectorize=True))
    np.testing.assert_allclose(out["final_cams"], g["final_cams"], rtol=0, atol=1e-8)
    assert abs(out["loss"] - float(g["loss"])) <= 1e-8 * abs(float(g["loss"]))
    for k in ("feat", "focal", "k1", "k2", "log_radius", "w_obs", "gt_cams", "w_strong", "w_reg"):
        ref_g = g["grad_" + k]
        np.testing.assert_allclose(out["grad_" + k].reshape(ref_g.shape), ref_g, rtol=5e-6, atol=5e-6 * np.abs(ref_g).max(), err_msg=k)


def test_camera_camera_between_gradients_through_the_reference_loop(ref):
    """Bundle adjustment with odometry (Between costs on consecutive cameras), backward_mode="implicit" through the REAL loop:
    the gradients w.r.t. the odometry measurements and weights too (thx_pg_vjp over the camera columns of the backward solve,
    ``_FusedAtbBA``) -- all eleven groups of tests/golden/ba_f64_camcam_implicit.npz (between.py:34-45 is plain autograd in the
    reference)."""
    th, thp = ref
    from tests.ba_common import run_ba_implicit

    class RefNames:
        Objective, SE3, Point3, Point2, Vector, Variable, Difference, Between = (th.Objective, th.SE3, th.Point3, th.Point2, th.Vector,
                                                                                th.Variable, th.Difference, th.Between)
        ScaleCostWeight, DiagonalCostWeight = th.ScaleCostWeight, th.DiagonalCostWeight
        RobustCostFunction, HuberLoss, WelschLoss = th.RobustCostFunction, th.HuberLoss, th.WelschLoss
        Reprojection = th.eb.Reprojection
        LevenbergMarquardt, TheseusLayer = th.LevenbergMarquardt, th.TheseusLayer
    g = load_golden("ba_f64_camcam_implicit")
    out = run_ba_implicit(RefNames, g, device=DEVICE,
                          opt_kwargs=dict(linear_solver_cls=thp.HipSchurSolver, linearization_kwargs=_kernels(), vectorize=True))
    np.testing.assert_allclose(out["final_cams"], g["final_cams"], rtol=0, atol=1e-7)
    assert abs(out["loss"] - float(g["loss"])) <= 1e-7 * abs(float(g["loss"]))
    for k in ("feat", "focal", "k1", "k2", "log_radius", "w_obs", "gt_cams", "w_strong", "w_reg", "cc_meas", "w_cc"):
        ref_g = g["grad_" + k]
        np.testing.assert_allclose(out["grad_" + k].reshape(ref_g.shape), ref_g, rtol=0, atol=5e-6 * max(np.abs(ref_g).max(), 1e-12),
                                   err_msg=k)


@pytest.mark.parametrize("name", ["ba_f64_unroll_lm", "ba_f64_camcam_unroll_lm", "ba_f64_trunc_conv_lm", "ba_f64_step_unroll_lm"])
def test_bundle_adjustment_unrolled_gradients_through_the_reference_loop(ref, name):
    """backward_mode "unroll" on a bundle-adjustment objective through the REAL loop with the Schur path behind it: every
    ``solve()`` is one autograd node over (cameras, points, auxiliary tensors) -- ``_FusedUnrolledSchurSolve`` (the call's Schur
    system rebuilt in the backward, thx_ba_unroll_vjp [+ thx_pg_unroll_vjp for the odometry]); retraction and error evaluation in
    between are the reference's own differentiable ops.  Gradients against the reference's own dense run."""
    th, thp = ref
    from tests.ba_common import run_ba_implicit
    from tests.test_unrolled_host import check_ba_unrolled

    class RefNames:
        Objective, SE3, Point3, Point2, Vector, Variable, Difference, Between = (th.Objective, th.SE3, th.Point3, th.Point2, th.Vector,
                                                                                th.Variable, th.Difference, th.Between)
        ScaleCostWeight, DiagonalCostWeight = th.ScaleCostWeight, th.DiagonalCostWeight
        RobustCostFunction, HuberLoss, WelschLoss = th.RobustCostFunction, th.HuberLoss, th.WelschLoss
        Reprojection = th.eb.Reprojection
        LevenbergMarquardt, TheseusLayer = th.LevenbergMarquardt, th.TheseusLayer
    g = load_golden(name)
    out = run_ba_implicit(RefNames, g, device=DEVICE,
                          opt_kwargs=dict(linear_solver_cls=thp.HipSchurSolver, linearization_kwargs=_kernels(), vectorize=True))
    for k in list(out):
        if k.startswith("grad_"):
            out[k] = out[k].reshape(g[k].shape)
    check_ba_unrolled(g, out)


def test_dogleg_on_bundle_adjustment_through_the_plugin(ref):
    """th.Dogleg needs ``linearization.Av``: for bundle adjustment the plugin answers from thx_ba_av (per-cost Jacobian blocks,
    no dense Jacobian).  Av against the reference's DenseLinearization under the same variable ordering, then the REAL
    th.Dogleg with HipSchurSolver against th.Dogleg on the reference's dense path: same trajectory and trust-region radii."""
    th, thp = ref
    from tests.ba_common import build_ba_objective

    class RefNames:
        Objective, SE3, Point3, Point2, Vector, Difference = th.Objective, th.SE3, th.Point3, th.Point2, th.Vector, th.Difference
        ScaleCostWeight, RobustCostFunction, HuberLoss, WelschLoss = th.ScaleCostWeight, th.RobustCostFunction, th.HuberLoss, th.WelschLoss
        Reprojection = th.eb.Reprojection
    g = load_golden("ba_f64_lm")
    out = {}
    for tag in ("plugin", "reference"):
        obj, cam_v, pt_v = build_ba_objective(RefNames, g, DEVICE)
        if DEVICE != "cpu":
            obj.to(DEVICE)
        obj.update()
        kw = dict(linear_solver_cls=thp.HipSchurSolver, linearization_kwargs=_kernels()) if tag == "plugin" else dict(
            linear_solver_cls=th.CholeskyDenseSolver)
        opt = th.Dogleg(obj, vectorize=True, abs_err_tolerance=0.0, rel_err_tolerance=0.0, max_iterations=4, step_size=1.0, **kw)
        lin = opt.linear_solver.linearization
        if tag == "plugin":
            lin.linearize()
            ref_lin = th.optimizer.DenseLinearization(obj, ordering=lin.ordering)
            ref_lin.linearize()
            v = torch.randn(obj.batch_size, lin.num_cols, dtype=torch.float64, generator=torch.Generator().manual_seed(1)).to(DEVICE)
            want = ref_lin.Av(v).detach().cpu().numpy()
            np.testing.assert_allclose(lin.Av(v).detach().cpu().numpy(), want, rtol=0, atol=1e-9 * np.abs(want).max())
        with torch.no_grad():
            info = opt.optimize(track_err_history=True, trust_region_init=2.0)
        used = sorted(set(g["obs_pt"].tolist()))
        out[tag] = (torch.stack([v.tensor for v in cam_v], 1).cpu(), torch.stack([pt_v[i].tensor for i in used], 1).cpu(),
                    info.err_history.cpu(), opt._trust_region.view(-1).cpu())
    a, b = out["plugin"], out["reference"]
    np.testing.assert_allclose(a[0].numpy(), b[0].numpy(), rtol=0, atol=1e-7)
    np.testing.assert_allclose(a[1].numpy(), b[1].numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(a[2].numpy(), b[2].numpy(), rtol=1e-6)
    np.testing.assert_array_equal(a[3].numpy(), b[3].numpy())
    assert (b[2][:, -1] < b[2][:, 0]).all()


# ---- the third hook set (Objective vectorization callbacks) -------------------------------------------------
def _counting(standin, names):
    calls = {n: 0 for n in names}
    for name in names:
        def spy(*a, _f=getattr(standin, name), _n=name, **k):
            calls[_n] += 1
            return _f(*a, **k)
        setattr(standin, name, spy)
    return calls


@cpu_only
@pytest.mark.parametrize("name", ["pg_f64_lm_adaptive_rejects", "pg2_f64_lm"])
def test_objective_hooks_keep_the_reference_loop_off_aten(ref, name):
    """With theseus_amd.plugin the REAL loop's retract_vars_sequence / error_metric / update go through the kernels: the
    reference's vectorized torch Jacobian pass (Vectorize._vectorize, run by every Objective.update) never runs, its torch
    retraction never runs, and the trajectory is still the recorded one."""
    th, thp = ref
    from tests.oracle_kernels import OracleKernels
    g = load_golden(name)
    _, _, kw = golden_problem(g)
    obj, poses = _objective(th, g)
    kw.pop("gauss_newton", False)
    standin = OracleKernels()
    calls = _counting(standin, ["pg_assemble", "retract", "pg_jacobians", "chol_factor"])
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=thp.HipCholeskySolver, linearization_kwargs=dict(kernels=standin),
                                vectorize=True, abs_err_tolerance=0.0, rel_err_tolerance=0.0,
                                max_iterations=kw.pop("max_iterations"), step_size=kw.pop("step_size"))
    hooks = opt.linear_solver.linearization.hooks
    assert hooks is not None and obj.vectorized and obj._retract_method == hooks.retract
    ref_calls = {"run": 0, "retract": 0, "err_iter": 0}
    for key in ref_calls:
        def spy(*a, _f=hooks.ref[key], _k=key, **k):
            ref_calls[_k] += 1
            return _f(*a, **k)
        hooks.ref[key] = spy
    with torch.no_grad():
        info = opt.optimize(track_err_history=True, **kw)
    assert ref_calls == {"run": 0, "retract": 0, "err_iter": 0}, ref_calls
    solves = calls["chol_factor"]
    assert calls["pg_assemble"] == solves and calls["retract"] == solves and calls["pg_jacobians"] >= solves
    final = torch.stack([p.tensor for p in poses], 1).detach().cpu().numpy()
    from tests.test_gpu_lm import well_conditioned_steps
    ok = well_conditioned_steps(g, g["delta"].shape[0])
    slack = 2.0 * (np.abs(g["delta"]).max(axis=2) * ~ok).sum(axis=0)
    assert (np.abs(final - g["final"]).reshape(final.shape[0], -1).max(1) <= 5e-8 + slack).all()
    k = min(info.err_history.shape[1], g["err_history"].shape[1])
    np.testing.assert_allclose(info.err_history[:, :k].detach().cpu().numpy(), g["err_history"][:, :k], rtol=2e-5)
    # Objective.error() / error_metric() as a user calls them: the (B, m) weighted error vector in cost add order
    ref_obj, _ = _objective(th, g)
    ref_obj.update({p.name: p.tensor for p in poses})
    np.testing.assert_allclose(obj.error().detach().cpu().numpy(), ref_obj.error().detach().cpu().numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(obj.error_metric().detach().cpu().numpy(), ref_obj.error_metric().detach().cpu().numpy(), rtol=1e-10)
    # somebody else's Linearization on the hooked objective is served Jacobians through the reference's own wrappers
    other = th.optimizer.DenseLinearization(obj)
    other.linearize()
    mine = opt.linear_solver.linearization
    mine.linearize()
    np.testing.assert_allclose(other.AtA.detach().cpu().numpy(), mine.AtA.detach().cpu().numpy(), rtol=1e-9, atol=1e-9)
    # disable_vectorization() (objective.py:945-951) takes the hooks out again
    obj.disable_vectorization()
    assert not obj.vectorized and obj._retract_method == th.Objective._retract_base


@cpu_only
def test_hooks_can_be_left_out(ref):
    th, thp = ref
    from tests.oracle_kernels import OracleKernels
    g = load_golden("pg_f64_lm")
    obj, poses = _objective(th, g)
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=thp.HipCholeskySolver, vectorize=True, max_iterations=6,
                                linearization_kwargs=dict(kernels=OracleKernels(), objective_hooks=False),
                                abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    assert opt.linear_solver.linearization.hooks is None and type(obj._vectorization_run.__self__).__name__ == "Vectorize"
    with torch.no_grad():
        opt.optimize(damping=1e-3)
    np.testing.assert_allclose(torch.stack([p.tensor for p in poses], 1).detach().cpu().numpy(), g["final"], rtol=0, atol=5e-8)


# ---- the reference's knobs ----------------------------------------------------------------------------------------------------
def test_global_params_of_the_reference_reach_the_kernels(ref):
    """torchlie.set_global_params / theseus.set_global_params (torchlie/global_params.py:61-68, theseus/global_params.py:
    63-80) are read at every launch once the plugin is imported."""
    th, thp = ref
    import torchlie
    import theseus_amd.kernels as tk
    try:
        torchlie.set_global_params({"so3_near_zero_eps_float64": 0.25, "so3_near_pi_eps_float32": 0.5})
        th.set_global_params({"se2_near_zero_eps_float64": 0.125})
        e = tk.lie_eps(torch.float64)
        assert (e.near_zero, e.d_near_zero, e.near_pi) == (0.25, 1e-2, 1e-7)
        assert tk.lie_eps(torch.float32).near_pi == 0.5 and tk.se2_eps(torch.float64).near_zero == 0.125
    finally:
        torchlie.reset_global_params()
        th.global_params._THESEUS_GLOBAL_PARAMS.reset()
    assert tk.lie_eps(torch.float64).near_zero == 5e-3 and tk.se2_eps(torch.float64).near_zero == 1e-6
    # a threshold change moves the result exactly as it moves the reference's: force the Taylor branch of log everywhere
    from tests.oracle_kernels import OracleKernels  # noqa: F401  (the stand-in reads oracle.lie.EPS, not these: host-side check only)


@cpu_only
def test_fast_approx_local_jacobians_is_honoured_through_the_generic_path(ref):
    """theseus/embodied/misc/local_cost_fn.py:43-57.  Set before construction: the plugin takes the generic block path, where
    the reference evaluates its own (identity) Jacobians; toggled afterwards on a fused linearization: a loud error."""
    th, thp = ref
    from tests.oracle_kernels import OracleKernels
    g = load_golden("pg_f64_lm")
    try:
        th.set_global_params({"fast_approx_local_jacobians": True})
        obj, _ = _objective(th, g)
        lin = thp.HipLinearization(obj, kernels=OracleKernels())
        assert not lin.fused
        lin.linearize()
        ref_lin = th.optimizer.DenseLinearization(obj)
        ref_lin.linearize()
        np.testing.assert_allclose(lin.AtA.detach().cpu().numpy(), ref_lin.AtA.detach().cpu().numpy(), rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(lin.Atb.detach().cpu().numpy(), ref_lin.Atb.detach().cpu().numpy(), rtol=1e-10, atol=1e-9)
        th.set_global_params({"fast_approx_local_jacobians": False})
        exact = th.optimizer.DenseLinearization(obj)
        exact.linearize()
        assert not np.allclose(exact.AtA.detach().cpu().numpy(), ref_lin.AtA.detach().cpu().numpy(), rtol=1e-6)   # the option does change the Hessian
        obj2, _ = _objective(th, g)
        fused = thp.HipLinearization(obj2, kernels=OracleKernels())
        assert fused.fused
        th.set_global_params({"fast_approx_local_jacobians": True})
        with pytest.raises(NotImplementedError, match="fast_approx_local_jacobians"):
            fused.linearize()
    finally:
        th.global_params._THESEUS_GLOBAL_PARAMS.reset()


@cpu_only
def test_check_singular_matches_the_reference(ref):
    """dense_solver.py:91-114: batch items whose (undamped) AtA is singular get an all-zero step and a RuntimeWarning, the
    others are solved."""
    th, thp = ref
    from tests.oracle_kernels import OracleKernels
    # fp32: the reference's check_singular path only runs in fp32 (its output buffer is torch.zeros(...) of the default dtype,
    # dense_solver.py:95, and the index_put of an fp64 solution into it raises)
    g = dict(load_golden("pg_f32_lm"))
    B = g["poses0"].shape[0]
    g["w_between"] = np.repeat(g["w_between"], B, axis=0).copy()
    g["w_prior"] = np.repeat(g["w_prior"], B, axis=0).copy()
    g["w_between"][1] = 0.0          # item 1: every cost weight zero -> A = 0 -> singular AtA
    g["w_prior"][1] = 0.0
    out = {}
    for tag in ("ref", "ours"):
        obj, _ = _objective(th, g)
        if tag == "ref":
            solver = th.CholeskyDenseSolver(obj, check_singular=True)
        else:
            solver = thp.HipCholeskySolver(obj, linearization_kwargs=_kernels(), check_singular=True)
        solver.linearization.linearize()
        with pytest.warns(RuntimeWarning, match="Singular matrix found in batch"):
            out[tag] = solver.solve(damping=0.1, ellipsoidal_damping=False).double()
    assert (out["ours"][1] == 0).all() and (out["ref"][1] == 0).all()
    good = [0, 2]
    np.testing.assert_allclose(out["ours"][good].detach().cpu().numpy(), out["ref"][good].detach().cpu().numpy(), rtol=0,
                               atol=2e-3 * np.abs(out["ref"][good].detach().cpu().numpy()).max())


@cpu_only
@pytest.mark.parametrize("tag", ["gn_unroll", "gn_trunc", "lm_unroll", "lm_trunc", "gn_trunc_conv"])
def test_reference_loop_differentiates_through_the_plugin_on_generic_objectives(ref, tag):
    """backward_mode "unroll" (TheseusLayer's default) / "truncated" with the REAL loop and the plugin's Linearization /
    LinearSolver on a generic objective: the Hessian is part of the graph (its dense form by torch from the reference's
    differentiable blocks), the damped solve runs on the kernels as one autograd node.  Gradients equal the ones the reference
    produced with its own dense path (tests/golden/simple_example.npz: u_* entries)."""
    th, thp = ref
    from tests.simple_example_common import UNROLLED
    g = load_golden("simple_example")
    _, cls, mode, okw, tol = next(u for u in UNROLLED if u[0] == tag)
    dt = torch.float64
    xl = torch.from_numpy(g["v_x"]).clone().requires_grad_(True)
    yl = torch.from_numpy(g["v_y"]).clone().requires_grad_(True)
    wl = torch.linspace(0.5, 1.5, 12, dtype=dt).view(1, -1).clone().requires_grad_(True)
    a, b = th.Vector(1, name="a", dtype=dt), th.Vector(1, name="b", dtype=dt)

    def f(optim_vars, aux_vars):
        return aux_vars[1].tensor - optim_vars[0].tensor * torch.exp(optim_vars[1].tensor * aux_vars[0].tensor)
    obj = th.Objective(dtype=dt)
    obj.add(th.AutoDiffCostFunction([a, b], f, 12, aux_vars=[th.Variable(xl, name="x"), th.Variable(yl, name="y")],
                                    cost_weight=th.DiagonalCostWeight(th.Variable(wl, name="w"))))
    opt = getattr(th, cls)(obj, max_iterations=int(g[f"u_{tag}_iters"]), abs_err_tolerance=tol, rel_err_tolerance=tol,
                           linear_solver_cls=thp.HipCholeskySolver, linearization_kwargs=_kernels())
    sol, info = th.TheseusLayer(opt).forward(input_tensors={"a": torch.ones(6, 1, dtype=dt), "b": 2.5 * torch.ones(6, 1, dtype=dt)},
                                             optimizer_kwargs=dict(track_err_history=True, backward_mode=mode, **okw))
    assert not opt.linear_solver.linearization.fused
    loss = ((sol["a"] - 0.5) ** 2).mean() + ((sol["b"] - 1.0) ** 2).mean()
    loss.backward()
    r = lambda k: g[f"u_{tag}_{k}"]  # noqa: E731
    np.testing.assert_allclose(sol["a"].detach().numpy(), r("a"), rtol=1e-10)
    assert abs(loss.item() - float(r("loss"))) < 1e-12
    for leaf, key in ((xl, "gx"), (yl, "gy"), (wl, "gw")):
        want = r(key)
        np.testing.assert_allclose(leaf.grad.numpy(), want, rtol=0, atol=1e-9 * np.abs(want).max(), err_msg=key)
    np.testing.assert_allclose(info.err_history.numpy(), r("err"), rtol=1e-6)


@pytest.mark.parametrize("tag", ["gn_unroll", "lm_unroll", "lm_trunc", "lm_ellips_unroll", "gn_trunc_conv", "lm_welsch_unroll",
                                 "lm_step_unroll"])
def test_reference_loop_differentiates_through_the_plugin_on_se3_pose_graphs(ref, tag):
    _unrolled_se3_through_the_reference_loop(ref, tag, "HipCholeskySolver")


@pytest.mark.parametrize("tag", ["gn_unroll", "lm_ellips_unroll", "gn_trunc_conv", "lm_welsch_unroll"])
def test_reference_loop_differentiates_through_the_tile_sparse_solver(ref, tag):
    """... and with ``HipSparseCholeskySolver`` behind the loop (its solve() used to fall through to the plain solve
    under backward_mode "unroll" -- a delta without a grad_fn on the GPU, i.e. silently incomplete gradients): the same node, the
    backward's solve runs along the tile pattern on a copy of the iteration's tile-packed factor, the columns follow the solver's
    reverse Cuthill-McKee ordering."""
    _unrolled_se3_through_the_reference_loop(ref, tag, "HipSparseCholeskySolver")


def _unrolled_se3_through_the_reference_loop(ref, tag, solver_name, mode_override=None, extra_okw=None):
    """backward_mode "unroll" / "truncated" on an SE3 pose graph through the REAL loop with the FUSED path behind it: the
    reference linearizes with the Hessian in the graph (nonlinear_least_squares.py:100-135); the plugin assembles H, g with the
    kernels and makes solve() one autograd node over the packed poses and the auxiliary tensors (_FusedUnrolledSolve:
    thx_pg_unroll_vjp + a copy of each iteration's factor); retraction and error evaluation in between are the reference's own
    differentiable ops.  Gradients against the reference's own dense run (tests/golden/pg_f64_unrolled.npz)."""
    import ast
    th, thp = ref
    g = load_golden("pg_f64_unrolled")
    dtype = torch.float64
    t = lambda a: torch.from_numpy(a).to(DEVICE)  # noqa: E731
    kw = ast.literal_eval(str(g[f"{tag}_kwargs"]))
    mode, iters, gn = kw.pop("mode"), kw.pop("max_iterations"), kw.pop("gauss_newton")
    if mode_override is not None:
        mode = mode_override
        kw.pop("backward_num_iterations", None) if mode == "unroll" else None
    P = int(g["P"])
    meas, wb = t(g["meas"]).requires_grad_(True), t(g["w_between"]).requires_grad_(True)
    tgt, wp = t(g["prior_target"]).requires_grad_(True), t(g["w_prior"])[:, :, :1].clone().requires_grad_(True)
    obj = th.Objective(dtype=dtype)
    p0 = t(g["poses0"]).clone().requires_grad_(f"{tag}_grad_poses0" in g)   # (UNROLL: the gradient reaches the initial values too)
    poses = [th.SE3(tensor=p0[:, k] if p0.requires_grad else p0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
    robust = f"{tag}_robust" in g      # (Welsch RobustCostFunction on every Between cost, one learnable log_loss_radius)
    lr = t(g[f"{tag}_log_radius"]).clone().requires_grad_(True) if robust else None
    radius = th.Vector(tensor=lr, name="log_loss_radius") if robust else None
    for k in range(g["edges"].shape[0]):
        i, j = g["edges"][k].tolist()
        cf = th.Between(poses[i], poses[j], th.SE3(tensor=meas[:, k], name=f"meas_{k}"),
                        th.DiagonalCostWeight(th.Variable(wb[:, k], name=f"w_{k}")), name=f"between_{k}")
        obj.add(th.RobustCostFunction(cf, th.WelschLoss, radius, name=f"robust_between_{k}") if robust else cf)
    for k in range(g["prior_idx"].shape[0]):
        obj.add(th.Difference(poses[int(g["prior_idx"][k])], th.SE3(tensor=tgt[:, k], name=f"prior_target_{k}"),
                              th.ScaleCostWeight(th.Variable(wp[:, k], name=f"pw_{k}")), name=f"prior_{k}"))
    cls = th.GaussNewton if gn else th.LevenbergMarquardt
    plug = (dict(linear_solver_cls=getattr(thp, solver_name), linearization_cls=thp.HipLinearization, linearization_kwargs=_kernels())
            if solver_name is not None else dict(linear_solver_cls=th.CholeskyDenseSolver))   # (None: the reference on its own)
    opt = cls(obj, **plug,
              vectorize=True, max_iterations=iters, step_size=float(g[f"{tag}_step"]) if f"{tag}_step" in g else 1.0,
              abs_err_tolerance=0.0, rel_err_tolerance=float(g[f"{tag}_rel_tol"]) if f"{tag}_rel_tol" in g else 0.0)
    assert solver_name is None or opt.linear_solver.linearization.fused
    layer = th.TheseusLayer(opt)
    if DEVICE != "cpu":
        layer.to(DEVICE)
    sol, info = layer.forward(optimizer_kwargs=dict(backward_mode=mode, track_err_history=True, **kw, **(extra_okw or {})))
    final = torch.stack([sol[f"pose_{k}"] for k in range(P)], 1)
    loss = (t(g["coef"]) * final).sum()
    loss.backward()
    if mode_override is not None or extra_okw:
        return info, final.detach(), meas.grad     # (a variant of the fixture's run: the caller compares)
    tol_x = 2e-8 if robust else 1e-9      # (tests/unrolled_common.py says why)
    np.testing.assert_allclose(final.detach().cpu().numpy(), g[f"{tag}_final"], rtol=0, atol=tol_x)
    assert abs(float(loss.detach()) - float(g[f"{tag}_loss"])) < 10 * tol_x
    for leaf, key in ((meas, "meas"), (wb, "w_between"), (tgt, "prior_target"), (wp, "w_prior")) + (((lr, "log_radius"),) if robust else ()):
        want = g[f"{tag}_grad_{key}"]
        np.testing.assert_allclose(leaf.grad.cpu().numpy(), want, rtol=0, atol=2e-6 * np.abs(want).max(), err_msg=key)
    if p0.requires_grad:
        want = g[f"{tag}_grad_poses0"]
        np.testing.assert_allclose(p0.grad.cpu().numpy(), want, rtol=0, atol=1e-5 * np.abs(want).max(), err_msg="poses0")
    if f"{tag}_conv" in g:
        assert info.converged_iter.tolist() == g[f"{tag}_conv"].tolist()


@cpu_only
@pytest.mark.parametrize("mode", ["unroll", "truncated"])
def test_bookkeeping_of_the_differentiated_iterations_matches_the_reference(ref, mode):
    """The iteration on which EVERY problem converges inside the differentiated iterations: the reference's
    _update_info runs before its convergence test (nonlinear_least_squares.py:192-203), so that iteration still competes for
    best_solution / best_err, and in UNROLL's single loop it is in err_history / state_history at [it + 1] with best_iter = it;
    TRUNCATED's _merge_infos (nonlinear_optimizer.py:220-266) copies only the counted columns and keeps the first loop's best_iter.
    theseus_amd's own loop (stand-in kernels) against the reference ON ITS OWN, same problem, tracking on."""
    th, _ = ref
    import theseus_amd as tha
    from tests.oracle_kernels import OracleKernels
    from tests.unrolled_common import run_pg_unrolled
    track = dict(track_best_solution=True, track_state_history=True)
    want, want_final, want_grad = _unrolled_se3_through_the_reference_loop(ref, "gn_trunc_conv", None, mode_override=mode,
                                                                           extra_okw=track)
    got, got_final, got_grad = run_pg_unrolled(tha, load_golden("pg_f64_unrolled"), "gn_trunc_conv", "cpu", OracleKernels(),
                                               mode_override=mode, extra_okw=track)
    np.testing.assert_allclose(got_final.numpy(), want_final.numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(got_grad.numpy(), want_grad.numpy(), rtol=0, atol=2e-6 * float(want_grad.abs().max()))
    eh_w, eh_g = want.err_history.numpy(), got.err_history.numpy()
    assert np.array_equal(np.isinf(eh_w), np.isinf(eh_g)), (eh_w, eh_g)          # the same columns were written
    np.testing.assert_allclose(eh_g[~np.isinf(eh_g)], eh_w[~np.isinf(eh_w)], rtol=1e-6)
    assert got.best_iter.tolist() == want.best_iter.tolist()
    np.testing.assert_allclose(got.best_err.detach().numpy(), want.best_err.detach().numpy(), rtol=1e-6)
    assert got.converged_iter.tolist() == want.converged_iter.tolist()
    assert [int(x.value) for x in got.status] == [int(x.value) for x in want.status]
    for name, hw in want.state_history.items():
        hg = got.state_history[name].numpy()
        assert np.array_equal(np.isinf(hw.numpy()), np.isinf(hg)), name
        fin = ~np.isinf(hg)
        np.testing.assert_allclose(hg[fin], hw.numpy()[fin], rtol=0, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(got.best_solution[name].numpy(), want.best_solution[name].numpy(), rtol=0, atol=1e-6)


@cpu_only
def test_generic_path_places_jacobian_blocks_like_dense_linearization(ref):
    """dense_linearization.py:44-52 writes Jacobian i of a cost into the columns that START at the cost's i-th variable and span
    ``J.shape[2]`` of them -- a list shorter than the variable list, a block wider than its variable, a later block OVERWRITING part
    of an earlier one, under a caller-supplied ordering: ``HipLinearization``'s A, b, AtA, Atb equal the reference's own
    DenseLinearization on the same objective (the reference's optimizer tests rely on the first two cases,
    tests/theseus_tests/optimizer/nonlinear/common.py:76-79)."""
    th, thp = ref
    from tests.oracle_kernels import OracleKernels
    B = 4
    gen = torch.Generator().manual_seed(5)

    class Odd(th.CostFunction):
        def __init__(self, vs, blocks, dim, name):
            super().__init__(th.ScaleCostWeight(torch.tensor(1.5)), name=name)
            for k, v in enumerate(vs):
                setattr(self, f"v{k}", v)
                self.register_optim_var(f"v{k}")
            self._blocks, self._dim = blocks, dim
            self._J = [torch.randn(B, dim, w, generator=gen) for w in blocks]
            self._e = torch.randn(B, dim, generator=gen)

        def error(self):
            return self._e

        def jacobians(self):
            return self._J, self._e

        def dim(self):
            return self._dim

        def _copy_impl(self, new_name=None):
            raise NotImplementedError

    a, b_, c, d = (th.Vector(n, name=nm) for n, nm in ((2, "a"), (1, "b"), (3, "c"), (2, "d")))
    for v in (a, b_, c, d):
        v.update(torch.zeros(B, v.dof()))
    obj = th.Objective()
    obj.add(Odd([a, b_, c], [6], 3, "wide"))            # one block over a, b and c (they are adjacent in the ordering below)
    obj.add(Odd([b_, c, d], [1, 3], 2, "short"))        # two blocks for three variables
    obj.add(Odd([a, b_], [3, 1], 4, "overlap"))         # the second block overwrites the last column of the first
    ordering = th.VariableOrdering(obj, default_order=False)
    for v in (d, a, b_, c):
        ordering.append(v)
    obj.update()
    ref_lin = th.DenseLinearization(obj, ordering=ordering)
    ref_lin.linearize()
    lin = thp.HipLinearization(obj, ordering=ordering, kernels=OracleKernels())
    assert not lin.fused
    lin.linearize()
    np.testing.assert_allclose(lin.A.numpy(), ref_lin.A.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(lin.b.numpy(), ref_lin.b.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(lin.AtA.numpy(), ref_lin.AtA.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lin.Atb.numpy(), ref_lin.Atb.numpy(), rtol=1e-5, atol=1e-5)
    v = torch.randn(B, lin.num_cols, generator=gen)
    np.testing.assert_allclose(lin.Av(v).numpy(), ref_lin.Av(v).numpy(), rtol=1e-5, atol=1e-5)


def test_dogleg_with_unrolled_gradients_is_refused_on_the_fused_path(ref):
    """th.Dogleg reads ``linearization.Av`` (dogleg.py:66), which the fused path answers from kernels outside autograd: with
    backward_mode="unroll" and something to differentiate the gradient would be silently incomplete -- refused loudly (the
    reference's loop re-raises as RuntimeError, nonlinear_least_squares.py:138-152); under no_grad the same call runs."""
    th, thp = ref
    g = load_golden("pg_f64_unrolled")
    t = lambda a: torch.from_numpy(a).to(DEVICE)  # noqa: E731
    meas = t(g["meas"]).clone().requires_grad_(True)
    obj = th.Objective(dtype=torch.float64)
    poses = [th.SE3(tensor=t(g["poses0"])[:, k].clone(), name=f"pose_{k}") for k in range(int(g["P"]))]
    for k in range(g["edges"].shape[0]):
        i, j = g["edges"][k].tolist()
        obj.add(th.Between(poses[i], poses[j], th.SE3(tensor=meas[:, k], name=f"meas_{k}"),
                           th.DiagonalCostWeight(th.Variable(t(g["w_between"])[:, k].clone(), name=f"w_{k}")), name=f"between_{k}"))
    obj.add(th.Difference(poses[int(g["prior_idx"][0])], th.SE3(tensor=t(g["prior_target"])[:, 0].clone(), name="target"),
                          th.ScaleCostWeight(th.Variable(t(g["w_prior"])[:, 0, :1].clone(), name="pw")), name="prior"))
    opt = th.Dogleg(obj, max_iterations=2, step_size=1.0, linear_solver_cls=thp.HipCholeskySolver,
                    linearization_cls=thp.HipLinearization, linearization_kwargs=_kernels(), vectorize=True)
    layer = th.TheseusLayer(opt)
    if DEVICE != "cpu":
        layer.to(DEVICE)
    with pytest.raises(RuntimeError, match="outside autograd"):
        layer.forward(optimizer_kwargs=dict(backward_mode="unroll"))
    with torch.no_grad():
        _, info = layer.forward(optimizer_kwargs=dict(backward_mode="unroll"))
    assert info.status is not None


@pytest.mark.parametrize("tag", ["gn_unroll", "lm_trunc", "lm_ellips_unroll"])
@pytest.mark.parametrize("fixture", ["pg2_f64_unrolled", "pg3_f64_unrolled"])
def test_reference_loop_differentiates_through_the_plugin_on_se2_and_so3_pose_graphs(ref, fixture, tag):
    """... and on the 3-dof groups (thx_pg2_unroll_vjp / thx_pgso3_unroll_vjp behind ``_FusedUnrolledSolve``)."""
    import ast
    th, thp = ref
    g = load_golden(fixture)
    G = th.SE2 if str(g["group"]) == "SE2" else th.SO3
    t = lambda a: torch.from_numpy(a).to(DEVICE)  # noqa: E731
    kw = ast.literal_eval(str(g[f"{tag}_kwargs"]))
    mode, iters, gn = kw.pop("mode"), kw.pop("max_iterations"), kw.pop("gauss_newton")
    P = int(g["P"])
    meas, wb = t(g["meas"]).requires_grad_(True), t(g["w_between"]).requires_grad_(True)
    tgt, wp = t(g["prior_target"]).requires_grad_(True), t(g["w_prior"])[:, :, :1].clone().requires_grad_(True)
    obj = th.Objective(dtype=torch.float64)
    poses = [G(tensor=t(g["poses0"])[:, k].clone(), name=f"pose_{k}") for k in range(P)]
    for k in range(g["edges"].shape[0]):
        i, j = g["edges"][k].tolist()
        obj.add(th.Between(poses[i], poses[j], G(tensor=meas[:, k], name=f"meas_{k}"),
                           th.DiagonalCostWeight(th.Variable(wb[:, k], name=f"w_{k}")), name=f"between_{k}"))
    for k in range(g["prior_idx"].shape[0]):
        obj.add(th.Difference(poses[int(g["prior_idx"][k])], G(tensor=tgt[:, k], name=f"tgt_{k}"),
                              th.ScaleCostWeight(th.Variable(wp[:, k], name=f"pw_{k}")), name=f"prior_{k}"))
    cls = th.GaussNewton if gn else th.LevenbergMarquardt
    opt = cls(obj, linear_solver_cls=thp.HipCholeskySolver, linearization_cls=thp.HipLinearization, linearization_kwargs=_kernels(),
              vectorize=True, max_iterations=iters, step_size=1.0, abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    assert opt.linear_solver.linearization.fused
    layer = th.TheseusLayer(opt)
    if DEVICE != "cpu":
        layer.to(DEVICE)
    sol, info = layer.forward(optimizer_kwargs=dict(backward_mode=mode, track_err_history=True, **kw))
    final = torch.stack([sol[f"pose_{k}"] for k in range(P)], 1)
    loss = (t(g["coef"]) * final).sum()
    loss.backward()
    np.testing.assert_allclose(final.detach().cpu().numpy(), g[f"{tag}_final"], rtol=0, atol=1e-9)
    for leaf, key in ((meas, "meas"), (wb, "w_between"), (tgt, "prior_target"), (wp, "w_prior")):
        want = g[f"{tag}_grad_{key}"]
        np.testing.assert_allclose(leaf.grad.cpu().numpy(), want, rtol=0, atol=2e-6 * np.abs(want).max(), err_msg=key)
