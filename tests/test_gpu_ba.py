"""-m gpu: bundle adjustment on the HIP kernels (csrc/ba_kernels.hip + the dense Cholesky on the Schur complement) against
the reference's DenseLinearization + CholeskyDenseSolver runs (tests/golden/ba_*.npz) and the oracle (oracle/ba.py)."""
import numpy as np
import pytest
import torch

from oracle import pose_graph as opg
from tests.ba_common import build_ba_objective, reference_columns, run_ba
from tests.helpers import ba_problem, f32_thresholds, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["ba_f64_lm", "ba_f64_gn", "ba_f32_lm"])
def test_ba_first_linear_system_matches_reference(name):
    """thx_ba_assemble + thx_ba_schur + Cholesky + thx_ba_backsub = the delta the reference's dense solve produced at the
    first iteration; g, diag(H) and the error metric against the reference's A^T b, diag(A^T A), error."""
    import theseus_amd as th
    g = load_golden(name)
    f32 = name.startswith("ba_f32")
    obj, _, _ = build_ba_objective(th, g, "cuda")
    opt = th.LevenbergMarquardt(obj, max_iterations=1)
    solver, lin = opt.linear_solver, opt.linear_solver.linearization
    obj.update()
    lin.linearize()
    cols, _ = reference_columns(g)
    Atb, AtA = g["Atb"][0][..., 0], g["AtA"][0]
    dref = np.diagonal(AtA, axis1=1, axis2=2)
    got_g, got_d = lin.g.cpu().numpy()[:, cols], lin.diag.cpu().numpy()[:, cols]
    if f32:
        # the reference's fp32 evaluation is itself a noisy draw (7e-4 relative on single entries of A^T b here): compare
        # with the exact values of the same fp32 inputs and require us to be inside the reference's own band
        import dataclasses
        p, state0, _, _ = ba_problem(g)
        p64 = dataclasses.replace(p, **{f.name: getattr(p, f.name).double() for f in dataclasses.fields(p)
                                        if isinstance(getattr(p, f.name), torch.Tensor) and getattr(p, f.name).is_floating_point()})
        with f32_thresholds():
            A, b = p64.dense_linearize((state0[0].double(), state0[1].double()))
            H64, g64 = opg.hessian(A, b)
            e64 = p64.error_metric((state0[0].double(), state0[1].double())).numpy()
        g64, d64 = g64[..., 0].numpy(), H64.diagonal(dim1=1, dim2=2).numpy()
        for got, ref32, exact in ((got_g, Atb, g64), (got_d, dref, d64)):
            sc = np.abs(exact).max()
            assert np.abs(got - exact).max() <= 3e-7 * sc
            assert np.abs(got - ref32).max() <= np.abs(ref32 - exact).max() + 3e-7 * sc
        np.testing.assert_allclose(obj.error_metric().cpu().numpy(), e64, rtol=3e-7)
    else:
        np.testing.assert_allclose(got_g, Atb, rtol=0, atol=1e-11 * np.abs(Atb).max())
        np.testing.assert_allclose(got_d, dref, rtol=1e-10, atol=1e-11 * np.abs(dref).max())
        np.testing.assert_allclose(obj.error_metric().cpu().numpy(), g["err0"], rtol=1e-12)
    import ast
    kw = ast.literal_eval(str(g["opt_kwargs"]))
    if kw["gauss_newton"]:
        delta = solver.solve()
    else:
        B = Atb.shape[0]
        lam = torch.full((B,), kw["damping"], dtype=lin.g.dtype, device="cuda")
        delta = solver.solve(damping=lam, ellipsoidal_damping=kw.get("ellipsoidal_damping", False), damping_eps=1e-8)
    want = g["delta"][0]
    np.testing.assert_allclose(delta.cpu().numpy()[:, cols], want, rtol=0, atol=(2e-3 if f32 else 1e-8) * max(1.0, np.abs(want).max()))


@pytest.mark.parametrize("name,tol_c,tol_p", [("ba_f64_lm", 1e-6, 1e-5), ("ba_f64_gn", 1e-6, 1e-5)])
def test_ba_trajectory_matches_reference(name, tol_c, tol_p):
    import theseus_amd as th
    g = load_golden(name)
    cams, pts, used, deltas, info, _ = run_ba(th, g, None, "cuda")
    np.testing.assert_allclose(cams.cpu().numpy(), g["final_cams"], rtol=0, atol=tol_c)
    np.testing.assert_allclose(pts.cpu().numpy(), g["final_pts"][:, used], rtol=0, atol=tol_p)
    k = min(info.err_history.shape[1], g["err_history"].shape[1])
    np.testing.assert_allclose(info.err_history[:, :k].numpy(), g["err_history"][:, :k], rtol=1e-6)
    assert all(s == th.NonlinearOptimizerStatus.MAX_ITERATIONS for s in info.status)


def test_ba_fp32_inside_reference_band():
    """fp32: the HIP trajectory is no farther from the exact trajectory of the same fp32 problem (fp64 oracle, fp32
    thresholds) than the reference's own fp32 run."""
    import dataclasses
    import theseus_amd as th
    g = load_golden("ba_f32_lm")
    p, (c0, p0), kw, used = ba_problem(g)
    d = lambda x: None if x is None else x.double()  # noqa: E731
    p64 = dataclasses.replace(p, **{f.name: d(getattr(p, f.name)) for f in dataclasses.fields(p)
                                    if isinstance(getattr(p, f.name), torch.Tensor) and getattr(p, f.name).is_floating_point()})
    with f32_thresholds():
        (xc, xp), xinfo = opg.lm_optimize(p64, (c0.double(), p0.double()), abs_err_tolerance=0.0, rel_err_tolerance=0.0, **kw)
    cams, pts, used2, _, info, _ = run_ba(th, g, None, "cuda")
    assert used2 == used
    dev_c = (cams.cpu().double() - xc).abs().max().item()
    ref_c = (torch.from_numpy(g["final_cams"]).double() - xc).abs().max().item()
    dev_p = (pts.cpu().double() - xp).abs().max().item()
    ref_p = (torch.from_numpy(g["final_pts"][:, used]).double() - xp).abs().max().item()
    assert dev_c <= 1.5 * ref_c + 1e-5 and dev_p <= 1.5 * ref_p + 1e-4, (dev_c, ref_c, dev_p, ref_p)
    hx = torch.stack(xinfo.err_history, 1)
    rel = ((info.err_history.double() - hx).abs() / hx).max().item()
    rel_ref = ((torch.from_numpy(g["err_history"]).double() - hx).abs() / hx).max().item()
    assert rel <= 1.5 * rel_ref + 1e-5, (rel, rel_ref)


def test_ba_non_positive_definite_sets_fail_status():
    import warnings
    import theseus_amd as th
    g = dict(load_golden("ba_f64_gn"))
    g["w_cam_prior"] = g["w_cam_prior"] * 0.0
    g["w_pt_prior"] = g["w_pt_prior"] * 0.0
    g["feat"] = g["feat"] * 0.0   # keeps shapes; the gauge freedom makes the undamped system singular
    obj, _, _ = build_ba_objective(th, g, "cuda")
    opt = th.GaussNewton(obj, max_iterations=2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        info = opt.optimize()
    assert all(s in (th.NonlinearOptimizerStatus.FAIL, th.NonlinearOptimizerStatus.MAX_ITERATIONS) for s in info.status)


# ---- multi-tile / full-size fixtures (oracle/gen_golden.py: ba_mid_*, ba_full_f64_lm) --------------------------------------
def _ba_first_system(th, g, f32=False, solver_kwargs=None):
    import ast
    obj, _, _ = build_ba_objective(th, g, "cuda")
    opt = th.LevenbergMarquardt(obj, max_iterations=1, linear_solver_kwargs=solver_kwargs)
    solver, lin = opt.linear_solver, opt.linear_solver.linearization
    obj.update()
    lin.linearize()
    cols, _ = reference_columns(g)
    Atb = g["Atb"][0][..., 0]
    np.testing.assert_allclose(lin.g.cpu().numpy()[:, cols], Atb, rtol=0, atol=(2e-4 if f32 else 1e-11) * np.abs(Atb).max())
    np.testing.assert_allclose(obj.error_metric().cpu().numpy(), g["err0"], rtol=1e-5 if f32 else 1e-12)
    kw = ast.literal_eval(str(g["opt_kwargs"]))
    lam = torch.full((Atb.shape[0],), kw["damping"], dtype=lin.g.dtype, device="cuda")
    delta = solver.solve(damping=lam, ellipsoidal_damping=kw.get("ellipsoidal_damping", False), damping_eps=1e-8)
    want = g["delta"][0]
    return np.abs(delta.cpu().numpy()[:, cols] - want).max() / max(1.0, np.abs(want).max()), solver


def test_ba_multi_tile_matches_reference():
    """32 cameras / 471 observed points / 2048 observations: the Schur tables span many block rows and the reduced camera
    system (192 x 192) takes two Cholesky tiles; first linear system, the whole adaptive ellipsoidal LM trajectory and the
    solution against the reference's dense run."""
    import theseus_amd as th
    g = load_golden("ba_mid_f64_lm")
    err, solver = _ba_first_system(th, g)
    assert solver.linearization.packed.nc == 192 and err <= 1e-8, err
    cams, pts, used, deltas, info, _ = run_ba(th, g, None, "cuda")
    dc = np.abs(cams.cpu().numpy() - g["final_cams"]).max()
    dp = np.abs(pts.cpu().numpy() - g["final_pts"][:, used]).max()
    print(f"[BA 32 cams] max |cam - reference| = {dc:.2e}, max |point - reference| = {dp:.2e}")
    assert dc <= 1e-6 and dp <= 1e-5, (dc, dp)
    k = min(info.err_history.shape[1], g["err_history"].shape[1])
    np.testing.assert_allclose(info.err_history[:, :k].numpy(), g["err_history"][:, :k], rtol=1e-6)
    cols, _ = reference_columns(g)
    if len(deltas) == g["delta"].shape[0]:
        for it, d in enumerate(deltas):
            np.testing.assert_allclose(d.cpu().numpy()[:, cols], g["delta"][it], rtol=0, atol=1e-6 * max(1.0, np.abs(g["delta"][it]).max()))


def test_ba_multi_tile_fp32_inside_reference_band():
    import dataclasses
    import theseus_amd as th
    g = load_golden("ba_mid_f32_lm")
    p, (c0, p0), kw, used = ba_problem(g)
    d = lambda x: None if x is None else x.double()  # noqa: E731
    p64 = dataclasses.replace(p, **{f.name: d(getattr(p, f.name)) for f in dataclasses.fields(p)
                                    if isinstance(getattr(p, f.name), torch.Tensor) and getattr(p, f.name).is_floating_point()})
    with f32_thresholds():
        (xc, xp), xinfo = opg.lm_optimize(p64, (c0.double(), p0.double()), abs_err_tolerance=0.0, rel_err_tolerance=0.0, **kw)
    cams, pts, used2, _, info, _ = run_ba(th, g, None, "cuda")
    dev_c = (cams.cpu().double() - xc).abs().max().item()
    ref_c = (torch.from_numpy(g["final_cams"]).double() - xc).abs().max().item()
    dev_p = (pts.cpu().double() - xp).abs().max().item()
    ref_p = (torch.from_numpy(g["final_pts"][:, used]).double() - xp).abs().max().item()
    print(f"[BA 32 cams fp32] |cam - exact|: hip {dev_c:.2e}, reference {ref_c:.2e}; |point - exact|: hip {dev_p:.2e}, reference {ref_p:.2e}")
    assert dev_c <= 1.5 * ref_c + 1e-5 and dev_p <= 1.5 * ref_p + 1e-4, (dev_c, ref_c, dev_p, ref_p)
    hx = torch.stack(xinfo.err_history, 1)
    rel = ((info.err_history.double() - hx).abs() / hx).max().item()
    rel_ref = ((torch.from_numpy(g["err_history"]).double() - hx).abs() / hx).max().item()
    assert rel <= 1.5 * rel_ref + 1e-5, (rel, rel_ref)


def test_ba_full_size_matches_the_reference_run():
    """BASELINE.json configs[3] at FULL size -- 512 SE3 cameras / 8192 Point3 (7481 observed) / 32768 robust Reprojection
    costs, one problem, fp64 -- against the REAL reference's DenseLinearization + CholeskyDenseSolver run (dense A 20.6 GB,
    A^T A 6.1 GB: oracle/gen_golden.py case ba_full_f64_lm): A^T b, the first linear solve, two adaptive ellipsoidal LM
    iterations, the solution.  The reduced camera system is 3072 x 3072 (24 Cholesky tiles)."""
    import theseus_amd as th
    g = load_golden("ba_full_f64_lm")
    assert int(g["C"]) == 512 and g["obs_cam"].shape[0] == 32768
    err, solver = _ba_first_system(th, g)
    assert solver.linearization.packed.nc == 3072 and solver.levels and solver.pattern.tree_levels < solver.pattern.ntiles and err <= 1e-7, err
    cams, pts, used, deltas, info, _ = run_ba(th, g, None, "cuda")
    dc = np.abs(cams.cpu().numpy() - g["final_cams"]).max()
    dp = np.abs(pts.cpu().numpy() - g["final_pts"][:, used]).max()
    print(f"[BA 512 cams] first solve: max |delta - reference| / |delta| = {err:.2e}; after 2 LM iterations: "
          f"max |cam - reference| = {dc:.2e}, max |point - reference| = {dp:.2e}")
    assert dc <= 1e-5 and dp <= 1e-5, (dc, dp)      # north_star's 1e-5, on a 27648-column problem
    k = min(info.err_history.shape[1], g["err_history"].shape[1])
    np.testing.assert_allclose(info.err_history[:, :k].numpy(), g["err_history"][:, :k], rtol=1e-6)


@pytest.mark.parametrize("name", ["ba_f64_implicit", "ba_f64_flatten_implicit"])
def test_ba_implicit_backward_matches_reference_gradients(name):
    """(``ba_f64_flatten_implicit``: the Reprojection costs wrapped with flatten_dims=True -- every image coordinate its own
    Huber term, robust_cost_function.py:89-96,118-133.)
    backward_mode="implicit" on a bundle-adjustment objective through the HIP kernels (thx_se3_retract_vjp -> solve with the
    cached Schur factor -> thx_ba_vjp): the gradients the REAL reference produced (oracle/gen_golden.py:gen_ba_implicit) w.r.t.
    log_loss_radius, the image features, the calibration (dual numbers), the observation weight, the strong camera priors'
    targets / weight and the regularisers' weight."""
    import theseus_amd as th
    from tests.ba_common import run_ba_implicit
    g = load_golden(name)
    got = run_ba_implicit(th, g, None, "cuda")
    np.testing.assert_allclose(got["final_cams"], g["final_cams"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(got["final_pts"], g["final_pts"], rtol=0, atol=1e-6)
    assert abs(got["loss"] - float(g["loss"])) < 1e-5
    for k in ("log_radius", "feat", "focal", "k1", "k2", "w_obs", "gt_cams", "w_strong", "w_reg"):
        want = g["grad_" + k]
        np.testing.assert_allclose(got["grad_" + k], want, rtol=0, atol=5e-6 * max(np.abs(want).max(), 1e-12), err_msg=k)


def test_ba_multi_tile_implicit_backward_matches_reference_gradients():
    """The same at 32 cameras / 471 observed points / 2048 observations (oracle/gen_golden.py: ba_mid_f64_implicit): the
    cached Schur factor spans two Cholesky tiles, the Schur tables many block rows."""
    import theseus_amd as th
    from tests.ba_common import run_ba_implicit
    g = load_golden("ba_mid_f64_implicit")
    assert int(g["C"]) == 32
    got = run_ba_implicit(th, g, None, "cuda")
    np.testing.assert_allclose(got["final_cams"], g["final_cams"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got["final_pts"], g["final_pts"], rtol=0, atol=1e-5)
    assert abs(got["loss"] - float(g["loss"])) < 1e-5 * max(1.0, abs(float(g["loss"])))
    for k in ("log_radius", "feat", "focal", "k1", "k2", "w_obs", "gt_cams", "w_strong", "w_reg"):
        want = g["grad_" + k]
        d = np.abs(got["grad_" + k] - want).max() / max(np.abs(want).max(), 1e-12)
        print(f"[BA 32 cams implicit] grad {k}: {d:.2e}")
        assert d <= 2e-5, (k, d)


def test_ba_full_size_fp32_inside_the_reference_band():
    """fp32 at the size tools / bench.py measure (512 cameras / 8192 points / 32768 observations): the reference has no fp32 run
    at this size (its dense A is 10 GB in fp32), so the band is the fp64 reference solution's: after the same two adaptive LM
    iterations the fp32 HIP trajectory's cost history agrees to fp32 level and the cameras / points stay within the tolerance
    the 32-camera fp32 test measured against the reference's own fp32 run (scaled by nothing: absolute)."""
    import theseus_amd as th
    g = load_golden("ba_full_f64_lm")
    g32 = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 else v) for k, v in g.items()}
    cams, pts, used, _, info, _ = run_ba(th, g32, None, "cuda")
    assert cams.dtype == torch.float32
    dc = np.abs(cams.cpu().double().numpy() - g["final_cams"]).max()
    dp = np.abs(pts.cpu().double().numpy() - g["final_pts"][:, used]).max()
    k = min(info.err_history.shape[1], g["err_history"].shape[1])
    rel = np.abs(info.err_history[:, :k].double().numpy() / g["err_history"][:, :k] - 1).max()
    print(f"[BA 512 cams fp32] max |cam - fp64 reference| = {dc:.2e}, max |point - fp64 reference| = {dp:.2e}, cost history rel {rel:.2e}")
    assert rel <= 2e-4 and dc <= 5e-3 and dp <= 5e-2, (rel, dc, dp)


@pytest.mark.parametrize("name", ["ba_f64_lm", "ba_f32_lm"])
def test_ba_av_and_dogleg_on_the_gpu(name):
    """thx_ba_av (Linearization.Av for the block linearization: what th.Dogleg / TrustRegion read) against the CPU stand-in built
    on the oracle's Jacobian blocks, and theseus_amd.Dogleg on a bundle-adjustment objective: GPU run == host run."""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    g = load_golden(name)
    f64 = g["cams0"].dtype == np.float64
    res = {}
    # (the host leg runs in fp64 on the same data: the stand-in kernels do not model the fp32 path's fp64 block buffers)
    g64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in g.items()}
    for dev in ("cuda", "cpu"):
        obj, cam_v, pt_v = build_ba_objective(th, g if dev == "cuda" else g64, dev)
        opt = th.Dogleg(obj, max_iterations=4, step_size=1.0, abs_err_tolerance=0.0, rel_err_tolerance=0.0,
                        linearization_kwargs=dict(kernels=OracleKernels()) if dev == "cpu" else None)
        lin = opt.linear_solver.linearization
        lin.linearize()
        v = torch.randn(lin.g.shape[0], lin.n, dtype=torch.float64, generator=torch.Generator().manual_seed(2)).to(lin.g.dtype).to(dev)
        av = lin.Av(v).cpu().double()
        assert av.shape == (lin.g.shape[0], lin.packed.m)
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(track_err_history=True, trust_region_init=2.0))
        res[dev] = (av, torch.stack([sol[v_.name] for v_ in cam_v], 1).cpu().double(), info.err_history.double(),
                    opt._trust_region.view(-1).cpu().double())
    a, b = res["cuda"], res["cpu"]
    scale = float(b[0].abs().max())
    if f64:
        np.testing.assert_allclose(a[0].numpy(), b[0].numpy(), rtol=0, atol=1e-10 * scale)
    else:   # fp32 against the fp64 evaluation: rounding, except for the few observations that sit at the Huber kink (the
            # rescale switches branch between the two precisions: a relative step of the size of the residual's rounding)
        diff = (a[0] - b[0]).abs().numpy()
        assert (diff > 2e-5 * scale).mean() < 0.01 and diff.max() < 1e-2 * scale, (diff.max(), scale)
    np.testing.assert_allclose(a[1].numpy(), b[1].numpy(), rtol=0, atol=1e-7 if f64 else 5e-3)
    np.testing.assert_allclose(a[2].numpy(), b[2].numpy(), rtol=1e-6 if f64 else 5e-3)
    if f64:
        np.testing.assert_array_equal(a[3].numpy(), b[3].numpy())
    assert (a[2][:, -1] < a[2][:, 0]).all()


# ---- round 6: the reduced camera system as a block list, factorised along its elimination tree ------------------------------
@pytest.mark.parametrize("name", ["ba_mid_f64_lm", "ba_mid_f32_lm"])
def test_schur_block_list_equals_the_dense_frame(name):
    """thx_ba_schur_blocks writes the SAME numbers as thx_ba_schur (same kernels, same arithmetic, another store): block c =
    S_cc, block C + k = the k-th camera pair's block -- transposed where the solver's order puts c2 behind c1 -- and the same
    rhs / Hinv / tvec; through the C ABI."""
    import theseus_amd as th
    g = load_golden(name)
    got = {}
    for ordering in ("natural", "nd"):
        obj, _, _ = build_ba_objective(th, g, "cuda")
        opt = th.LevenbergMarquardt(obj, max_iterations=1, linear_solver_kwargs=dict(ordering=ordering))
        solver, lin = opt.linear_solver, opt.linear_solver.linearization
        obj.update()
        lin.linearize()
        lam = torch.full((lin.g.shape[0],), 0.02, dtype=lin.g.dtype, device="cuda")
        delta = solver.solve(damping=lam, ellipsoidal_damping=True, damping_eps=1e-8).clone()
        got[ordering] = (solver, delta)
    dense, lev = got["natural"][0], got["nd"][0]
    assert not dense.levels and lev.levels and lev.Sc is not None and dense.S is not None and lev.S is None
    s = lev.linearization.packed.structure
    C, K = s.num_cams, s.num_blocks
    Sc = lev.Sc[:, :36 * (C + K)].view(-1, C + K, 6, 6)
    S = dense.S
    dst = lev._level_t["blk_dst"].cpu().numpy()
    c1, c2 = s.t["blk_c1"][:K].astype(np.int64), s.t["blk_c2"][:K].astype(np.int64)
    for c in range(C):
        assert torch.equal(Sc[:, c], S[:, 6 * c:6 * c + 6, 6 * c:6 * c + 6])
    flipped = 0
    for k in range(K):
        blk = S[:, 6 * c1[k]:6 * c1[k] + 6, 6 * c2[k]:6 * c2[k] + 6]
        tr = bool((dst[k] >> 30) & 1)
        flipped += tr
        assert (dst[k] & 0x3fffffff) == C + k
        assert torch.equal(Sc[:, C + k], blk.transpose(1, 2) if tr else blk)
    assert flipped > 0                                                   # (the dissection really reorders cameras)
    assert torch.equal(lev.rhs, dense.rhs) and torch.equal(lev.Hinv, dense.Hinv) and torch.equal(lev.tvec, dense.tvec)
    # the two factorisations (other order, other summation order) solve the same system
    d0, d1 = got["natural"][1], got["nd"][1]
    tol = 1e-9 if lev.Sc.dtype == torch.float64 else 2e-3
    assert ((d0 - d1).abs().max() / d0.abs().max()).item() < tol
    # the cached factor solves another right-hand side the same in both modes (the implicit backward's solve)
    r = torch.randn(lin.g.shape[0], lin.n, dtype=lin.g.dtype, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    w0, w1 = dense.solve_with_factor(r), lev.solve_with_factor(r)
    assert ((w0 - w1).abs().max() / w0.abs().max()).item() < tol


@pytest.mark.parametrize("name", ["ba_mid_f64_lm", "ba_mid_f32_lm"])
def test_dense_tiles_through_the_matrix_core_scatter_equal_the_gather(name):
    """A reduced camera system's tiles hold up to 21 x 21 blocks and take the LDS gather by default; forced through the matrix-core
    path (thx_chol_schedule.hb_scatter_max_pieces: hb_add's overflow chunks of 64 pieces, several per tile) the level-scheduled
    factorisation must give the same bits -- both ways add the same values to the same sums."""
    import theseus_amd as th
    g = load_golden(name)
    got = []
    for limit in (-1, 1 << 20):
        obj, _, _ = build_ba_objective(th, g, "cuda")
        opt = th.LevenbergMarquardt(obj, max_iterations=1, linear_solver_kwargs=dict(ordering="nd"))
        solver, lin = opt.linear_solver, opt.linear_solver.linearization
        obj.update()
        lin.linearize()
        prev = solver.K.chol_hb_scatter_max_pieces(limit)
        try:
            lam = torch.full((lin.g.shape[0],), 0.02, dtype=lin.g.dtype, device="cuda")
            got.append(solver.solve(damping=lam, ellipsoidal_damping=True, damping_eps=1e-8).clone())
        finally:
            solver.K.chol_hb_scatter_max_pieces(prev)
        assert solver.levels
        pieces = solver._level_layout.c.max_tile_pieces
    assert pieces > 64, pieces     # (the default is the gather here)
    assert torch.equal(got[0], got[1])


@pytest.mark.parametrize("ordering", ["natural", "nd", "md"])
def test_ba_full_size_first_solve_under_every_camera_order(ordering):
    """The reference's first linear solve at 512 / 8192 / 32768 (tests/golden/ba_full_f64_lm) under the cameras' own order
    (dense frame, column by column), a tile-level nested dissection and minimum degree (block list, level schedule)."""
    import theseus_amd as th
    g = load_golden("ba_full_f64_lm")
    err, solver = _ba_first_system(th, g, solver_kwargs=dict(ordering=ordering))
    assert solver.levels == (ordering != "natural") and err <= 1e-7, err
    if solver.levels:
        assert solver.pattern.ntiles == 25 and int(solver.info.abs().sum()) == 0


def test_ba_level_mode_not_positive_definite_is_reported():
    import warnings
    import theseus_amd as th
    g = dict(load_golden("ba_mid_f64_lm"))
    g["w_cam_prior"] = g["w_cam_prior"] * 0.0
    g["w_pt_prior"] = g["w_pt_prior"] * 0.0
    g["feat"] = g["feat"] * 0.0   # the gauge freedom makes the undamped system singular
    obj, _, _ = build_ba_objective(th, g, "cuda")
    opt = th.GaussNewton(obj, max_iterations=2, linear_solver_kwargs=dict(ordering="nd"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        info = opt.optimize()
    assert opt.linear_solver.levels
    assert all(s in (th.NonlinearOptimizerStatus.FAIL, th.NonlinearOptimizerStatus.MAX_ITERATIONS) for s in info.status)
