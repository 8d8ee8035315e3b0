"""CPU twin (TEST stand-in kernels) of the fused unrolled backward of SE3 pose graphs: the host logic of
theseus_amd/autograd.py:PGUnrolledIteration -- the chain over iterations, the retraction's two paths, the factor copies, the fixed
order sum of the poses' gradients, accept / reject selects -- against the REAL reference's gradients.  The kernel's maths itself is
checked without a GPU by tests/test_unroll_math_host.py, the kernel on the GPU by tests/test_gpu_unrolled.py."""
import numpy as np
import pytest
import torch

from tests.helpers import load_golden
from tests.unrolled_common import run_pg_unrolled


@pytest.mark.parametrize("tag", ["gn_unroll", "lm_unroll", "lm_trunc", "lm_ellips_unroll", "gn_trunc_conv", "lm_welsch_unroll",
                                 "gn_huberflat_trunc", "lm_step_unroll"])
def test_differentiating_through_the_iterations_of_a_pose_graph(tag):
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    run_pg_unrolled(th, load_golden("pg_f64_unrolled"), tag, "cpu", OracleKernels())


@pytest.mark.parametrize("tag", ["gn_unroll", "lm_ellips_unroll", "gn_trunc_conv", "lm_welsch_unroll"])
def test_differentiating_through_the_iterations_with_the_tile_sparse_solver(tag):
    """The same node over ``HipSparseCholeskySolver`` (reverse Cuthill-McKee column order, the backward's solve along the tile
    pattern on a copy of the iteration's factor: ``solve_with_snapshot``)."""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    run_pg_unrolled(th, load_golden("pg_f64_unrolled"), tag, "cpu", OracleKernels(), solver_cls=th.HipSparseCholeskySolver)


@pytest.mark.parametrize("tag", ["gn_unroll", "lm_trunc", "lm_ellips_unroll"])
@pytest.mark.parametrize("fixture", ["pg2_f64_unrolled", "pg3_f64_unrolled"])
def test_differentiating_through_the_iterations_of_se2_and_so3_pose_graphs(fixture, tag):
    """The 3-dof groups (thx_pg2_unroll_vjp / thx_pgso3_unroll_vjp; SE2: plain autograd everywhere, SO3: torchlie's conventions) on
    the problems of the implicit fixtures, differentiated through the reference's iterations (oracle/gen_golden.py:
    gen_pg23_unrolled)."""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    run_pg_unrolled(th, load_golden(fixture), tag, "cpu", OracleKernels())


BA_UNROLLED = ["ba_f64_unroll_lm", "ba_f64_flatten_trunc_lm", "ba_f64_camcam_unroll_lm", "ba_f64_trunc_conv_lm", "ba_f64_step_unroll_lm"]


def check_ba_unrolled(g, got, grad_tol=5e-6):
    np.testing.assert_allclose(got["final_cams"], g["final_cams"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(got["final_pts"], g["final_pts"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got["err_history"], g["err_history"], rtol=1e-6)
    assert abs(got["loss"] - float(g["loss"])) < 1e-5
    if "rel_tol" in g and float(g["rel_tol"]) > 0:   # convergence tests on: who converged, and when
        np.testing.assert_array_equal(got["converged_iter"], g["converged_iter"])
        np.testing.assert_array_equal(got["status"], g["status"])
    keys = ("log_radius", "feat", "focal", "k1", "k2", "w_obs", "gt_cams", "w_strong", "w_reg")
    if "cc_edges" in g:
        keys += ("cc_meas", "w_cc")
    for k in keys:
        want = g["grad_" + k]
        np.testing.assert_allclose(got["grad_" + k], want, rtol=0, atol=grad_tol * max(np.abs(want).max(), 1e-12), err_msg=k)
    for k in ("cams0", "pts0"):    # UNROLL: the INITIAL values of the optimisation variables (raw entries; tests/unrolled_common.py)
        if "grad_" + k in g:
            want = g["grad_" + k]
            np.testing.assert_allclose(got["grad_" + k], want, rtol=0, atol=1e-5 * np.abs(want).max(), err_msg=k)


@pytest.mark.parametrize("name", BA_UNROLLED)
def test_bundle_adjustment_unrolled_gradients_match_the_reference(name):
    """BackwardMode.UNROLL / TRUNCATED on a bundle-adjustment objective (theseus_amd/ba.py:BAUnrolledIteration; TEST stand-in
    kernels here, the HIP kernels in tests/test_gpu_unrolled.py) against the gradients the REAL reference produced
    (oracle/gen_golden.py:gen_ba_implicit with mode="unroll" / "truncated"): adaptive LM with ellipsoidal damping through all
    iterations; flatten_dims Huber + spherical damping through the last two of four; camera-camera Between costs next to the
    reprojections; the convergence tests on (ba_f64_trunc_conv_lm: the three problems converge, and are frozen, at iterations 3, 4, 5
    of which the last five are differentiated).  Gradients w.r.t. log_loss_radius, the image features, the calibration, the observation weight, the strong
    camera priors' targets / weight, the regularisers' weight (+ the odometry measurements / weights)."""
    import theseus_amd as th
    from tests.ba_common import run_ba_implicit
    from tests.oracle_kernels import OracleKernels
    g = load_golden(name)
    check_ba_unrolled(g, run_ba_implicit(th, g, OracleKernels(), "cpu"))


def test_best_solution_and_state_history_are_tracked_through_differentiated_iterations():
    """track_best_solution / track_state_history together with backward_mode="unroll" (the reference's _update_info keeps detached
    copies, nonlinear_optimizer.py:150-207): same gradients as without tracking, the history's last slot is the solution, the best
    solution is the history's slot with the smallest error."""
    import torch
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    g = load_golden("pg_f64_unrolled")
    t = torch.from_numpy
    out = {}
    for track in (False, True):
        meas = t(g["meas"]).clone().requires_grad_(True)
        obj = th.Objective(dtype=torch.float64)
        P = int(g["P"])
        poses = [th.SE3(tensor=t(g["poses0"])[:, k].clone(), name=f"pose_{k}") for k in range(P)]
        for k in range(g["edges"].shape[0]):
            i, j = g["edges"][k].tolist()
            obj.add(th.Between(poses[i], poses[j], th.SE3(tensor=meas[:, k], name=f"meas_{k}"),
                               th.DiagonalCostWeight(th.Variable(t(g["w_between"])[:, k].clone(), name=f"w_{k}")), name=f"between_{k}"))
        obj.add(th.Difference(poses[int(g["prior_idx"][0])], th.SE3(tensor=t(g["prior_target"])[:, 0].clone(), name="target"),
                              th.ScaleCostWeight(th.Variable(t(g["w_prior"])[:, 0, :1].clone(), name="pw")), name="prior"))
        opt = th.LevenbergMarquardt(obj, max_iterations=4, abs_err_tolerance=0.0, rel_err_tolerance=0.0,
                                    linearization_kwargs=dict(kernels=OracleKernels()))
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(
            backward_mode="unroll", damping=0.05, adaptive_damping=True, track_err_history=True, track_best_solution=track,
            track_state_history=track))
        final = torch.stack([sol[f"pose_{k}"] for k in range(P)], 1)
        (t(g["coef"]) * final).sum().backward()
        out[track] = (final.detach(), meas.grad.clone(), info)
    np.testing.assert_array_equal(out[True][0].numpy(), out[False][0].numpy())
    np.testing.assert_array_equal(out[True][1].numpy(), out[False][1].numpy())
    info = out[True][2]
    hist = info.state_history["pose_3"]                                  # (B, 3, 4, K + 1), default dtype (fp32) like the reference's
    np.testing.assert_allclose(hist[..., -1].numpy(), out[True][0][:, 3].numpy(), rtol=0, atol=1e-6)
    errs = info.err_history
    k_best = errs.argmin(dim=1)
    np.testing.assert_allclose(info.best_err.numpy(), errs.min(dim=1).values.numpy(), rtol=1e-6)
    for b in range(errs.shape[0]):
        np.testing.assert_allclose(info.best_solution["pose_3"][b].numpy(), hist[b, ..., int(k_best[b])].numpy(), rtol=0, atol=1e-6)


def test_unrolled_gradients_of_a_dropped_singular_item_are_zero():
    """check_singular=True (dense_solver.py:91-103) under backward_mode="unroll": a batch item whose system is singular gets a zero
    step in every iteration -- its final poses are its initial ones, so NOTHING it depends on may receive a gradient through it
    (the reference's masked assignment), and the broken factor of that item must not leak NaNs into the others' gradients."""
    import warnings
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    g = dict(load_golden("pg_f64_lm_adaptive_ellips"))     # batched DiagonalCostWeights
    B, P = g["poses0"].shape[0], int(g["P"])
    wb, wp = g["w_between"].copy(), np.repeat(g["w_prior"], B, axis=0).copy()
    wb[2] = 0.0
    wp[2] = 0.0

    def run(items):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a[items]))  # noqa: E731
        leaves = dict(meas=t(g["meas"]).requires_grad_(True), w_between=t(wb).requires_grad_(True),
                      prior_target=t(g["prior_target"]).requires_grad_(True), w_prior=t(wp)[:, :, :1].clone().requires_grad_(True))
        poses0 = t(g["poses0"])
        obj = th.Objective(dtype=torch.float64)
        poses = [th.SE3(tensor=poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
        for k in range(g["edges"].shape[0]):
            i, j = g["edges"][k].tolist()
            obj.add(th.Between(poses[i], poses[j], th.SE3(tensor=leaves["meas"][:, k], name=f"meas_{k}"),
                               th.DiagonalCostWeight(th.Variable(leaves["w_between"][:, k], name=f"w_{k}")), name=f"between_{k}"))
        for k in range(g["prior_idx"].shape[0]):
            obj.add(th.Difference(poses[int(g["prior_idx"][k])], th.SE3(tensor=leaves["prior_target"][:, k], name=f"tgt_{k}"),
                                  th.ScaleCostWeight(th.Variable(leaves["w_prior"][:, k], name=f"pw_{k}")), name=f"prior_{k}"))
        opt = th.LevenbergMarquardt(obj, max_iterations=3, abs_err_tolerance=0.0, rel_err_tolerance=0.0,
                                    linearization_kwargs=dict(kernels=OracleKernels()), linear_solver_kwargs=dict(check_singular=True))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)      # "Singular matrix found in batch"
            sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(backward_mode="unroll", damping=0.1))
        final = torch.stack([sol[f"pose_{k}"] for k in range(P)], 1)
        (final * torch.linspace(0.5, 1.5, final[0].numel(), dtype=torch.float64).view(1, *final.shape[1:])).sum().backward()
        return final.detach(), poses0, {k: v.grad for k, v in leaves.items()}
    final, poses0, grads = run([0, 1, 2, 3])
    assert torch.equal(final[2], poses0[2])                                  # the dropped item never moved
    for k, gr in grads.items():
        assert torch.isfinite(gr).all(), k
        assert float(gr[2].abs().max()) == 0.0, k                            # ... and nothing reaches its inputs
    _, _, ref = run([0, 1, 3])                                               # the others: as if the dropped item were not there
    for k in grads:
        np.testing.assert_allclose(grads[k][[0, 1, 3]].numpy(), ref[k].numpy(), rtol=1e-9, atol=1e-12, err_msg=k)
