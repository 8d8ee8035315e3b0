"""-m gpu end-to-end parity: theseus_amd Objective/LevenbergMarquardt/TheseusLayer (HIP path)
against trajectories recorded from the REAL reference (tests/golden, oracle/gen_golden.py)."""
import numpy as np
import pytest
import torch

from tests.helpers import golden_problem, load_golden

pytestmark = pytest.mark.gpu


def build_objective(th, g, device="cuda"):
    """Same construction order as oracle/gen_golden.py:build_reference_objective."""
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    dtype = torch.from_numpy(g["poses0"]).dtype
    obj = th.Objective(dtype=dtype)
    P = int(g["P"])
    poses0 = t(g["poses0"])
    poses = [th.SE3(tensor=poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
    for k in range(g["edges"].shape[0]):
        i, j = g["edges"][k].tolist()
        cw = th.DiagonalCostWeight(th.Variable(t(g["w_between"])[:, k].clone(), name=f"w_{k}"))
        m = th.SE3(tensor=t(g["meas"])[:, k].clone(), name=f"meas_{k}")
        obj.add(th.Between(poses[i], poses[j], m, cw, name=f"between_{k}"))
    for k in range(g["prior_idx"].shape[0]):
        tgt = th.SE3(tensor=t(g["prior_target"])[:, k].clone(), name=f"prior_target_{k}")
        sw = th.ScaleCostWeight(th.Variable(t(g["w_prior"])[:, k, :1].clone(), name=f"pw_{k}"))
        obj.add(th.Difference(poses[int(g["prior_idx"][k])], tgt, sw, name=f"prior_{k}"))
    return obj, poses


CASES = [("pg_f64_lm", 2e-8), ("pg_f64_gn", 2e-8), ("pg_f64_lm_adaptive", 2e-8),
         ("pg_f64_lm_adaptive_ellips", 2e-8), ("pg_f64_lm_adaptive_rejects", 2e-8), ("pg_f32_lm", 5e-3)]


@pytest.mark.parametrize("name,tol", CASES)
def test_lm_trajectory_matches_reference(name, tol):
    import theseus_amd as th
    g = load_golden(name)
    _, _, kw = golden_problem(g)
    obj, poses = build_objective(th, g)
    gn = kw.pop("gauss_newton")
    okw = dict(max_iterations=kw.pop("max_iterations"), step_size=kw.pop("step_size"),
               abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    opt = (th.GaussNewton if gn else th.LevenbergMarquardt)(obj, linear_solver_cls=th.HipCholeskySolver, **okw)
    lin = opt.linear_solver.linearization
    # structure parity is bit exact (linearization.py:31-41)
    assert lin.var_start_cols == list(g["var_start_cols"]) and lin.var_dims == list(g["var_dims"])
    assert lin.num_rows == int(g["num_rows"]) and lin.num_cols == int(g["num_cols"])
    deltas = []
    layer = th.TheseusLayer(opt)
    sol, info = layer.forward(None, optimizer_kwargs=dict(track_err_history=True,
                                                          end_iter_callback=lambda o, i, d, it: deltas.append(d.clone()),
                                                          **kw))
    final = torch.stack([sol[f"pose_{k}"] for k in range(int(g["P"]))], 1).cpu().numpy()
    np.testing.assert_allclose(final, g["final"], rtol=0, atol=tol)
    if len(deltas) == g["delta"].shape[0]:
        for it, d in enumerate(deltas):
            np.testing.assert_allclose(d.cpu().numpy(), g["delta"][it], rtol=0,
                                       atol=tol * max(1.0, np.abs(g["delta"][it]).max()))
    k = min(info.err_history.shape[1], g["err_history"].shape[1])
    np.testing.assert_allclose(info.err_history[:, :k].numpy(), g["err_history"][:, :k],
                               rtol=2e-5 if tol < 1e-6 else 3e-3)
    assert all(s == th.NonlinearOptimizerStatus.MAX_ITERATIONS for s in info.status)


def test_first_linearization_properties_match_reference():
    import theseus_amd as th
    g = load_golden("pg_f64_lm")
    obj, _ = build_objective(th, g)
    opt = th.LevenbergMarquardt(obj, max_iterations=1)
    lin = opt.linear_solver.linearization
    obj.update()
    lin.linearize()
    sc = np.abs(g["AtA"][0]).max()
    np.testing.assert_allclose(lin.AtA.cpu().numpy(), g["AtA"][0], rtol=0, atol=sc * 5e-12)
    np.testing.assert_allclose(lin.Atb.cpu().numpy(), g["Atb"][0], rtol=0, atol=np.abs(g["Atb"][0]).max() * 5e-12)
    np.testing.assert_allclose(lin.A.cpu().numpy(), g["A0"], rtol=0, atol=np.abs(g["A0"]).max() * 1e-11)
    np.testing.assert_allclose(lin.b.cpu().numpy(), g["b0"], rtol=0, atol=np.abs(g["b0"]).max() * 1e-11)
    v = torch.randn(lin.AtA.shape[0], lin.num_cols, dtype=torch.float64, device="cuda")
    np.testing.assert_allclose(lin.Av(v).cpu().numpy(), (torch.from_numpy(g["A0"]) @ v.cpu().unsqueeze(2)).squeeze(2).numpy(),
                               rtol=1e-9, atol=1e-8)
    dref = torch.from_numpy(g["AtA"][0]).diagonal(dim1=1, dim2=2) * v.cpu()
    np.testing.assert_allclose(lin.diagonal_scaling(v).cpu().numpy(), dref.numpy(), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(obj.error().cpu().numpy(), -g["b0"], rtol=0, atol=np.abs(g["b0"]).max() * 1e-11)
    np.testing.assert_allclose(obj.error_metric().cpu().numpy(), g["err0"], rtol=1e-12)


def test_non_positive_definite_sets_fail_status():
    """An all-zero weight makes H singular: the reference raises inside solve -> FAIL status
    (nonlinear_least_squares.py:138-152)."""
    import warnings
    import theseus_amd as th
    g = load_golden("pg_f64_gn")
    g = dict(g)
    g["w_between"] = g["w_between"] * 0.0
    g["w_prior"] = g["w_prior"] * 0.0
    obj, _ = build_objective(th, g)
    opt = th.GaussNewton(obj, max_iterations=3)
    obj.update()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        info = opt.optimize()
    assert all(s == th.NonlinearOptimizerStatus.FAIL for s in info.status)
    assert any("not positive-definite" in str(x.message) for x in w)


def test_unsupported_objective_raises_instead_of_falling_back():
    import theseus_amd as th

    class Weird(th.Difference):
        pass

    class NotSupported(th.CostFunction):
        def __init__(self, v, w):
            super().__init__(w, None)
            self.v = v
        def optim_vars(self): return [self.v]
        def aux_vars(self): return self.weight.aux_vars()
        def error(self): return self.v.log_map()
        def dim(self): return 6

    obj = th.Objective(dtype=torch.float64)
    v = th.SE3(tensor=torch.eye(3, 4, dtype=torch.float64, device="cuda").view(1, 3, 4), name="v")
    obj.add(NotSupported(v, th.ScaleCostWeight(torch.tensor(1.0, dtype=torch.float64, device="cuda"))))
    with pytest.raises(NotImplementedError):
        th.LevenbergMarquardt(obj)
