"""-m gpu: BackwardMode.UNROLL / TRUNCATED through the HIP kernels -- the generic path (thx_block_assemble, the tiled Cholesky's
damped factorisation, thx_chol_solve with a copy of each iteration's factor in the backward), SE3 / SE2 / SO3 pose graphs
(thx_pg*_unroll_vjp) and bundle adjustment (thx_ba_unroll_vjp) -- against the REAL reference's gradients (tests/golden/), plus the
bundle-adjustment objective with camera-camera costs.  CPU twins with the stand-in kernels: tests/test_generic_host.py,
tests/test_unrolled_host.py; last run on the GPU: profiles/r4/j_pytest_gpu_unrolled_ba.txt."""
import pytest

from tests.helpers import load_golden
from tests.simple_example_common import run_unrolled

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["gn_unroll", "gn_trunc", "lm_unroll", "lm_trunc", "gn_trunc_conv"])
def test_differentiating_through_the_iterations_on_the_gpu(tag):
    import theseus_amd as th
    run_unrolled(th, load_golden("simple_example"), tag, "cuda")


@pytest.mark.parametrize("tag", ["gn_unroll", "lm_unroll", "lm_trunc", "lm_ellips_unroll", "gn_trunc_conv", "lm_welsch_unroll",
                                 "gn_huberflat_trunc", "lm_step_unroll"])
def test_differentiating_through_the_iterations_of_a_pose_graph_on_the_gpu(tag):
    """BackwardMode.UNROLL / TRUNCATED on an SE3 pose graph through the HIP kernels (thx_pg_unroll_vjp, thx_se3_retract_vjp,
    thx_chol_solve with a copy of each iteration's factor) against the REAL reference's gradients
    (tests/golden/pg_f64_unrolled.npz).  CPU twin: tests/test_unrolled_host.py; the kernel's maths on the host:
    tests/test_unroll_math_host.py."""
    import theseus_amd as th
    from tests.unrolled_common import run_pg_unrolled
    run_pg_unrolled(th, load_golden("pg_f64_unrolled"), tag, "cuda")


@pytest.mark.parametrize("tag", ["gn_unroll", "lm_ellips_unroll", "gn_trunc_conv", "lm_welsch_unroll"])
def test_differentiating_through_the_iterations_with_the_tile_sparse_solver_on_the_gpu(tag):
    """... over ``HipSparseCholeskySolver``: thx_chol_solve_sparse on a copy of the iteration's tile-packed factor in the backward."""
    import theseus_amd as th
    from tests.unrolled_common import run_pg_unrolled
    run_pg_unrolled(th, load_golden("pg_f64_unrolled"), tag, "cuda", solver_cls=th.HipSparseCholeskySolver)


@pytest.mark.parametrize("tag", ["gn_unroll", "lm_trunc", "lm_ellips_unroll"])
@pytest.mark.parametrize("fixture", ["pg2_f64_unrolled", "pg3_f64_unrolled"])
def test_differentiating_through_the_iterations_of_se2_and_so3_pose_graphs_on_the_gpu(fixture, tag):
    """thx_pg2_unroll_vjp / thx_pgso3_unroll_vjp against the REAL reference's gradients (CPU twin: tests/test_unrolled_host.py)."""
    import theseus_amd as th
    from tests.unrolled_common import run_pg_unrolled
    run_pg_unrolled(th, load_golden(fixture), tag, "cuda")


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_unroll_vjp_kernel_against_autograd_through_the_oracle(dtype):
    """thx_pg_unroll_vjp on its own: per-cost gradients of phi = -(J w).(r + J delta) for random w, delta against torch autograd
    through the oracle's Between / Local formulas (tests/oracle_kernels.py:pg_unroll_vjp): fp64 to rounding; fp32 storage
    (double arithmetic inside, the same fp32-rounded inputs on both sides, fp32 Taylor thresholds) to the output's rounding."""
    import contextlib
    import dataclasses
    import torch
    from tests.gpu_helpers import to_device_problem
    from tests.helpers import f32_thresholds, golden_problem
    from tests.oracle_kernels import OracleKernels
    from theseus_amd.kernels import default_kernels
    g = load_golden("pg_f64_unrolled")
    p, poses0, _ = golden_problem({**g, "opt_kwargs": "{}"})
    dt = torch.float64 if dtype == "f64" else torch.float32
    r = (lambda x: x) if dtype == "f64" else (lambda x: x.float().double())
    p = dataclasses.replace(p, meas=r(p.meas), w_between=r(p.w_between), prior_target=r(p.prior_target), w_prior=r(p.w_prior))
    poses0 = r(poses0)
    B, n = poses0.shape[0], 6 * p.num_poses
    gen = torch.Generator().manual_seed(3)
    w, d = (r(torch.randn(B, n, dtype=torch.float64, generator=gen)) for _ in range(2))
    cast = lambda x: x.to(dt)  # noqa: E731
    p_d = dataclasses.replace(p, meas=cast(p.meas), w_between=cast(p.w_between), prior_target=cast(p.prior_target), w_prior=cast(p.w_prior))
    s, t = to_device_problem(p_d, cast(poses0))
    s64, t64 = to_device_problem(p, poses0, device="cpu")
    E, Kp = s.num_edges, s.num_priors
    shapes = [(E, B, 3, 4), (E, B, 3, 4), (E, B, 3, 4), (E, B, 6), (Kp, B, 3, 4), (Kp, B, 3, 4), (Kp, B, 6)]
    got = [torch.zeros(*sh, dtype=dt, device="cuda") for sh in shapes]
    default_kernels().pg_unroll_vjp(s.on("cuda"), t, cast(w).cuda(), cast(d).cuda(), *got)
    want = [torch.zeros(*sh, dtype=torch.float64) for sh in shapes]
    with (f32_thresholds() if dtype == "f32" else contextlib.nullcontext()):
        OracleKernels().pg_unroll_vjp(s64.on("cpu"), t64, w, d, *want)
    tol = 1e-10 if dtype == "f64" else 5e-7
    for a, b_, name in zip(got, want, ("pose_i", "pose_j", "meas", "w_between", "pose_prior", "prior_target", "w_prior")):
        assert (a.cpu().double() - b_).abs().max() <= tol * max(1.0, float(b_.abs().max())), name


def test_ba_with_camera_camera_costs_on_the_gpu():
    """Bundle adjustment with Between (odometry) costs on consecutive cameras through the HIP kernels -- thx_pg_assemble /
    thx_pg_error over the camera buffer joined with the Schur complement (theseus_amd/ba.py) -- against the REAL reference's run
    (tests/golden/ba_f64_camcam_lm.npz).  CPU twin with the stand-in kernels: tests/test_ba_host.py."""
    import numpy as np
    import theseus_amd as th
    from tests.ba_common import reference_columns, run_ba
    g = load_golden("ba_f64_camcam_lm")
    cams, pts, used, deltas, info, opt = run_ba(th, g, None, "cuda")
    assert len(opt.linear_solver.linearization.packed.cc_costs) == g["cc_edges"].shape[0]
    np.testing.assert_allclose(cams.cpu().numpy(), g["final_cams"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(pts.cpu().numpy(), g["final_pts"][:, used], rtol=0, atol=1e-5)
    k = min(info.err_history.shape[1], g["err_history"].shape[1])
    np.testing.assert_allclose(info.err_history[:, :k].numpy(), g["err_history"][:, :k], rtol=1e-6)
    cols, _ = reference_columns(g)
    np.testing.assert_allclose(deltas[0].cpu().numpy()[:, cols], g["delta"][0], rtol=0, atol=1e-7 * max(1.0, np.abs(g["delta"][0]).max()))


def test_ba_with_camera_camera_costs_implicit_backward_on_the_gpu():
    """... and its implicit backward: all gradient groups of tests/golden/ba_f64_camcam_implicit.npz incl. the odometry measurements
    and weights (thx_pg_vjp over the camera columns of the backward solve)."""
    import numpy as np
    import theseus_amd as th
    from tests.ba_common import run_ba_implicit
    g = load_golden("ba_f64_camcam_implicit")
    got = run_ba_implicit(th, g, None, "cuda")
    np.testing.assert_allclose(got["final_cams"], g["final_cams"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(got["final_pts"], g["final_pts"], rtol=0, atol=1e-6)
    assert abs(got["loss"] - float(g["loss"])) < 1e-5
    for k in ("log_radius", "feat", "focal", "k1", "k2", "w_obs", "gt_cams", "w_strong", "w_reg", "cc_meas", "w_cc"):
        want = g["grad_" + k]
        np.testing.assert_allclose(got["grad_" + k], want, rtol=0, atol=5e-6 * max(np.abs(want).max(), 1e-12), err_msg=k)


@pytest.mark.parametrize("name", ["ba_f64_unroll_lm", "ba_f64_flatten_trunc_lm", "ba_f64_camcam_unroll_lm", "ba_f64_trunc_conv_lm",
                                  "ba_f64_step_unroll_lm"])
def test_bundle_adjustment_unrolled_gradients_on_the_gpu(name):
    """BackwardMode.UNROLL / TRUNCATED on a bundle-adjustment objective through the HIP kernels (thx_ba_unroll_vjp, the Schur system
    of every differentiated iteration rebuilt and solved in the backward, thx_pg_unroll_vjp for the camera-camera costs) against the
    REAL reference's gradients.  CPU twin: tests/test_unrolled_host.py; the kernel's maths on the host:
    tests/test_unroll_math_host.py."""
    import theseus_amd as th
    from tests.ba_common import run_ba_implicit
    from tests.test_unrolled_host import check_ba_unrolled
    g = load_golden(name)
    check_ba_unrolled(g, run_ba_implicit(th, g, None, "cuda"))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("ellipsoidal", [False, True])
def test_ba_unroll_vjp_kernel_against_autograd_through_the_oracle(dtype, ellipsoidal):
    """thx_ba_unroll_vjp on its own: the per-cost gradients for random w, delta (and lambda) against torch autograd through the
    oracle's Reprojection / Difference formulas (tests/oracle_kernels.py:ba_unroll_vjp) on the same (fp32-rounded) inputs."""
    import contextlib
    import numpy as np
    import torch
    import theseus_amd as th
    from tests.ba_common import build_ba_objective
    from tests.helpers import f32_thresholds
    from tests.oracle_kernels import OracleKernels
    from theseus_amd import _lib
    g = dict(load_golden("ba_f64_unroll_lm"))
    if dtype == "f32":
        g = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 and v.ndim > 0 else v) for k, v in g.items()}
    g64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in g.items()}
    packs = []
    for gg, dev, K in ((g, "cuda", None), (g64, "cpu", OracleKernels())):
        obj, _, _ = build_ba_objective(th, gg, dev)
        lin = th.HipSchurLinearization(obj, **({} if K is None else dict(kernels=K)))
        lin.packed.sync(force=True)
        packs.append(lin.packed)
    pd, pc = packs
    B, n = pd.batch, pd.n
    dt = torch.float64 if dtype == "f64" else torch.float32
    gen = torch.Generator().manual_seed(11)
    w, d = (torch.randn(B, n, dtype=torch.float64, generator=gen).to(dt) * 0.05 for _ in range(2))
    lam = (0.01 + torch.rand(B, dtype=torch.float64, generator=gen)).to(dt) if ellipsoidal else None
    s = pd.structure
    O, Kc, Kp = s.num_obs, s.num_cam_priors, s.num_pt_priors
    shapes = dict(cam_obs=(O, B, 3, 4), pt_obs=(O, B, 3), feat=(O, B, 2), w_obs=(O, B, 2), focal=(O, B), k1=(O, B), k2=(O, B),
                  log_radius_obs=(O, B, 1), cam_prior_cam=(Kc, B, 3, 4), cam_prior_target=(Kc, B, 3, 4), w_cam_prior=(Kc, B, 6),
                  pt_prior_pt=(Kp, B, 3), pt_prior_target=(Kp, B, 3), w_pt_prior=(Kp, B, 3))
    assert set(shapes) == set(_lib.BA_UNROLL_GRADS)
    got = {k: torch.zeros(*sh, dtype=dt, device="cuda") for k, sh in shapes.items()}
    want = {k: torch.zeros(*sh, dtype=torch.float64) for k, sh in shapes.items()}
    pd.K.ba_unroll_vjp(pd.dstruct, pd.tensors, w.cuda(), d.cuda(), got, ell_damping=None if lam is None else lam.cuda())
    with (f32_thresholds() if dtype == "f32" else contextlib.nullcontext()):
        pc.K.ba_unroll_vjp(pc.dstruct, pc.tensors, w.double(), d.double(), want, ell_damping=None if lam is None else lam.double())
    tol = 1e-10 if dtype == "f64" else 5e-7
    for k in shapes:
        a, b_ = got[k].cpu().double(), want[k]
        assert float(b_.abs().max()) > 0, k
        assert (a - b_).abs().max() <= tol * max(1.0, float(b_.abs().max())), (k, float((a - b_).abs().max()), float(b_.abs().max()))
