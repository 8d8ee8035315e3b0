"""Bundle adjustment through the host path on CPU (PackedBA, HipSchurLinearization / HipSchurSolver, the LM loop on a
(cameras, points) state) with the TEST stand-in kernels: reproduces the reference's DenseLinearization +
CholeskyDenseSolver trajectories.  GPU twin: tests/test_gpu_ba.py."""
import numpy as np
import pytest
import torch

from tests.ba_common import reference_columns, run_ba
from tests.helpers import load_golden


@pytest.mark.parametrize("name", ["ba_f64_lm", "ba_f64_gn", "ba_f64_camcam_lm"])
def test_ba_host_path_matches_reference(name):
    """(``ba_f64_camcam_lm``: Between costs on consecutive cameras next to the reprojections -- beyond the example's shape; their
    blocks come from the pose-graph kernels over the camera buffer and join the Schur complement, theseus_amd/ba.py.)"""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    g = load_golden(name)
    cams, pts, used, deltas, info, opt = run_ba(th, g, OracleKernels())
    np.testing.assert_allclose(cams.numpy(), g["final_cams"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(pts.numpy(), g["final_pts"][:, used], rtol=0, atol=1e-5)
    k = min(info.err_history.shape[1], g["err_history"].shape[1])
    np.testing.assert_allclose(info.err_history[:, :k].numpy(), g["err_history"][:, :k], rtol=1e-6)
    cols, _ = reference_columns(g)
    np.testing.assert_allclose(deltas[0].numpy()[:, cols], g["delta"][0], rtol=0, atol=1e-7 * max(1.0, np.abs(g["delta"][0]).max()))
    lin = opt.linear_solver.linearization
    assert lin.num_cols == int(g["num_cols"]) and lin.num_rows == int(g["num_rows"])


def test_schur_block_tables_partition_the_pair_list():
    """BAStructure's tables for thx_ba_schur: every off-diagonal pair (cam(o2) < cam(o1)) lies in exactly one block run
    [begin, end) of constant (c1, c2), the diagonal pairs close each camera's range, blocks are unique."""
    import numpy as np
    from theseus_amd.ba import BAStructure
    rng = np.random.default_rng(0)
    C, Np = 9, 40
    u = np.unique(np.stack([rng.integers(0, C, 160), rng.integers(0, Np, 160)], 1), axis=0)
    oc, op = u[:, 0], u[:, 1]
    s = BAStructure(C, Np, oc, op, np.arange(C), np.arange(Np))
    t = s.t
    pc1, pc2 = oc[t["pair_o1"]], t["pair_c2"]
    assert (op[t["pair_o1"]] == op[t["pair_o2"]]).all() and (pc2 <= pc1).all()
    runs = t["blk_ptr"].reshape(-1, 2)
    assert runs.shape[0] == s.num_blocks == len(set(zip(t["blk_c1"].tolist(), t["blk_c2"].tolist())))
    covered = np.zeros(s.num_pairs, bool)
    for k, (a, b) in enumerate(runs):
        assert b > a and (pc1[a:b] == t["blk_c1"][k]).all() and (pc2[a:b] == t["blk_c2"][k]).all()
        assert not covered[a:b].any()
        covered[a:b] = True
    assert (covered == (pc2 < pc1)).all()
    for c in range(C):
        a, b = t["pair_dptr"][c], t["pair_ptr"][c + 1]
        assert (pc1[a:b] == c).all() and (pc2[a:b] == c).all() and (pc2[t["pair_ptr"][c]:a] < c).all()
    # every pair of observations of a common point appears once (lower triangle incl. the diagonal)
    want = sum(k * (k + 1) // 2 for k in np.bincount(op, minlength=Np))
    assert s.num_pairs == want


@pytest.mark.parametrize("name", ["ba_f64_implicit", "ba_f64_flatten_implicit", "ba_f64_camcam_implicit"])
def test_ba_implicit_backward_matches_reference_gradients(name):
    """(``ba_f64_flatten_implicit``: the Reprojection costs wrapped with flatten_dims=True -- every image coordinate its own
    Huber term, robust_cost_function.py:89-96,118-133.)
    backward_mode="implicit" on a bundle-adjustment objective through theseus_amd's host path (BAImplicitStep: retract VJP ->
    solve with the cached Schur factor -> thx_ba_vjp; TEST stand-in kernels here, HIP kernels in tests/test_gpu_ba.py): the
    gradients the REAL reference produced (oracle/gen_golden.py:gen_ba_implicit) w.r.t. log_loss_radius, the image features,
    the calibration, the observation weight, the strong camera priors' targets / weight and the regularisers' weight."""
    import ast
    import numpy as np
    import torch
    import theseus_amd as th
    from tests.ba_common import run_ba_implicit
    from tests.helpers import load_golden
    from tests.oracle_kernels import OracleKernels
    g = load_golden(name)
    got = run_ba_implicit(th, g, OracleKernels(), "cpu")
    np.testing.assert_allclose(got["final_cams"], g["final_cams"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(got["final_pts"], g["final_pts"], rtol=0, atol=1e-6)
    assert abs(got["loss"] - float(g["loss"])) < 1e-5
    keys = ("log_radius", "feat", "focal", "k1", "k2", "w_obs", "gt_cams", "w_strong", "w_reg")
    if "cc_edges" in g:   # ba_f64_camcam_implicit: + the odometry measurements / weights (thx_pg_vjp over the camera columns)
        keys += ("cc_meas", "w_cc")
    for k in keys:
        want = g["grad_" + k]
        np.testing.assert_allclose(got["grad_" + k], want, rtol=0, atol=5e-6 * max(np.abs(want).max(), 1e-12), err_msg=k)


def test_banded_reduced_system_is_factorised_along_its_tile_pattern():
    """Cameras on a line with local tracks (the reference's generator): the reduced camera system S is banded, the Schur solver
    hands thx_chol_factor_sparse the tile pattern of its Cholesky factor (the stand-in asserts that the numeric fill stays inside
    it) and the steps equal those of the dense factorisation of the same S."""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    from theseus_amd.utils.synthetic_ba import make_ba_objective
    out = {}
    for sparse in (True, False):
        K = OracleKernels()
        calls = {"sparse": 0}
        fs = K.chol_factor_sparse   # (the stand-in factorises densely and checks the fill against the pattern)
        K.chol_factor_sparse = lambda *a, **k: (calls.__setitem__("sparse", calls["sparse"] + 1), fs(*a, **k))[1]
        obj, meta = make_ba_objective(96, 768, 2, track_length=4, dtype=torch.float64, device="cpu", seed=1, kernels=K)
        opt = th.LevenbergMarquardt(obj, max_iterations=2, abs_err_tolerance=0.0, rel_err_tolerance=0.0,
                                    linearization_kwargs=dict(kernels=K),
                                    linear_solver_kwargs=dict(sparse_reduced_system=sparse))
        solver = opt.linear_solver
        with torch.no_grad():
            info = opt.optimize(damping=1e-2, adaptive_damping=True, ellipsoidal_damping=True, track_err_history=True)
        pat = solver.pattern
        assert pat.ntiles == 5 and pat.l_tiles < 15            # 96 cameras = 576 columns: the far corner of S is empty
        assert solver.sparse == sparse and (calls["sparse"] > 0) == sparse
        out[sparse] = (solver.delta.clone(), info.err_history.clone())
    np.testing.assert_allclose(out[True][0].numpy(), out[False][0].numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(out[True][1].numpy(), out[False][1].numpy(), rtol=1e-12)
    assert (out[True][1][:, -1] < out[True][1][:, 0]).all()


def test_camera_camera_costs_av_rows_and_dogleg():
    """``Av`` of a bundle-adjustment linearization with camera-camera Between costs: the odometry rows (thx_pg_jacobians over the
    camera buffer) next to the thx_ba_av rows, against the oracle's dense A in the reference's row / column layout; Dogleg (which
    reads Av) runs on the objective."""
    import theseus_amd as th
    from tests.ba_common import build_ba_objective, reference_columns
    from tests.helpers import ba_problem
    from tests.oracle_kernels import OracleKernels
    g = load_golden("ba_f64_camcam_lm")
    obj, cam_v, pt_v = build_ba_objective(th, g, "cpu")
    opt = th.Dogleg(obj, max_iterations=3, abs_err_tolerance=0.0, rel_err_tolerance=0.0, linearization_kwargs=dict(kernels=OracleKernels()))
    lin = opt.linear_solver.linearization
    assert len(lin.packed.cc_costs) == g["cc_edges"].shape[0]
    lin.linearize()
    p, state0, _, used = ba_problem(g)
    A, b = p.dense_linearize(state0)                       # reference layout: rows in cost add order, columns in insertion order
    cols, _ = reference_columns(g)
    v = torch.randn(A.shape[0], lin.n, dtype=torch.float64, generator=torch.Generator().manual_seed(4))
    want = (A @ v[:, cols].unsqueeze(2)).squeeze(2)        # v is in the linearization's order (cameras, then points)
    np.testing.assert_allclose(lin.Av(v).numpy(), want.numpy(), rtol=0, atol=1e-9 * float(want.abs().max()))
    np.testing.assert_allclose(lin.g.numpy()[:, cols], (A.transpose(1, 2) @ b.unsqueeze(2)).squeeze(2).numpy(), rtol=0,
                               atol=1e-9 * float(lin.g.abs().max()))
    with torch.no_grad():
        _, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(track_err_history=True, trust_region_init=2.0))
    assert (info.err_history[:, -1] < info.err_history[:, 0]).all()


# ---- round 6: the reduced camera system as a block list in a dissection order of the cameras (level mode) -- the HOST side ----
class _LevelStandIns:
    """CPU stand-ins of the four kernels the level mode of HipSchurSolver adds (mixed into OracleKernels): thx_ba_schur_blocks as
    the dense-frame stand-in + a scatter by the solver's block tables (incl. the transposed blocks), thx_chol_factor_levels /
    thx_chol_solve_levels as a dense Cholesky of the matrix the PADDED piece tables describe (identity on the padding), and
    thx_vec_gather.  What they check is every table the host builds: block ids, transposition flags, piece tables, vector maps."""

    def ba_schur_blocks(self, s, Hcc, Hpp, W, g, damping, ellipsoidal, damping_eps, Sc, diag_blk, blk_dst, rhs, Hinv, tvec, info):
        h = s.host
        C, K = h.num_cams, h.num_blocks
        S = torch.zeros(g.shape[0], 6 * C, 6 * C, dtype=Sc.dtype)
        self.ba_schur(s, Hcc, Hpp, W, g, damping, ellipsoidal, damping_eps, S, rhs, Hinv, tvec, info)
        Sc.zero_()
        blocks = Sc[:, :36 * (C + K)].view(-1, C + K, 6, 6)
        for c in range(C):
            blocks[:, int(diag_blk[c])] = S[:, 6 * c:6 * c + 6, 6 * c:6 * c + 6]
        c1, c2 = h.t["blk_c1"][:K].astype("int64"), h.t["blk_c2"][:K].astype("int64")
        for k in range(K):
            d = int(blk_dst[k])
            blk = S[:, 6 * c1[k]:6 * c1[k] + 6, 6 * c2[k]:6 * c2[k] + 6]
            blocks[:, d & 0x3fffffff] = blk.transpose(1, 2) if (d >> 30) & 1 else blk

    def _padded_dense(self, layout, Hc, pattern):
        T, nt = 128, pattern.ntiles
        tile_ptr, piece_blk, piece_rc = (layout.t[k].numpy() for k in ("tile_ptr", "piece_blk", "piece_rc"))
        A = torch.zeros(Hc.shape[0], nt * T, nt * T, dtype=Hc.dtype)
        for ti in range(nt):
            for tj in range(ti + 1):
                t = ti * (ti + 1) // 2 + tj
                for pc in range(tile_ptr[t], tile_ptr[t + 1]):
                    rc = int(piece_rc[pc]) & 0xFFFFFFFF
                    r, c = rc >> 16, rc & 0xFFFF
                    blk = Hc[:, 36 * int(piece_blk[pc]):36 * int(piece_blk[pc]) + 36].view(-1, 6, 6)
                    A[:, T * ti + r:T * ti + r + 6, T * tj + c:T * tj + c + 6] = blk
        A = torch.tril(A) + torch.tril(A, -1).transpose(1, 2)
        pad = torch.from_numpy(pattern.col_of_pad < 0)
        A[:, pad, pad] = 1.0
        return A

    def chol_factor_levels(self, layout, Hc, damping, ellipsoidal, damping_eps, L, panels, info, pattern, rhs=None, y=None):
        assert damping is None
        Ld, inf = torch.linalg.cholesky_ex(self._padded_dense(layout, Hc, pattern))
        info.copy_(inf.to(info.dtype))
        self._Ld = Ld
        if rhs is not None:
            y.copy_(torch.linalg.solve_triangular(Ld, rhs.unsqueeze(2), upper=False).squeeze(2))

    def chol_solve_levels(self, L, panels, rhs, x, pattern, which=0):
        v = rhs.unsqueeze(2)
        if which in (0, 2):
            v = torch.linalg.solve_triangular(self._Ld, v, upper=False)
        if which in (0, 1):
            v = torch.linalg.solve_triangular(self._Ld.transpose(1, 2), v, upper=True)
        x.copy_(v.squeeze(2))

    def vec_gather(self, src, dst, idx):
        i = idx.long()
        dst.copy_(torch.where(i >= 0, src[:, i.clamp(min=0)], torch.zeros((), dtype=src.dtype)))


def test_level_mode_tables_give_the_dense_frames_solution():
    """96 cameras on a line (5 padded tiles, a real dissection: cameras change places, some camera pairs are stored transposed):
    HipSchurSolver(ordering="nd") through the stand-ins above takes the same LM steps as the natural order on the dense frame --
    and its cached-factor solve (the implicit backward's) the same as well."""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    from theseus_amd.utils.synthetic_ba import make_ba_objective

    class K2(_LevelStandIns, OracleKernels):
        pass
    out = {}
    for ordering in ("natural", "nd"):
        K = K2()
        obj, meta = make_ba_objective(96, 768, 2, track_length=4, dtype=torch.float64, device="cpu", seed=1, kernels=K)
        opt = th.LevenbergMarquardt(obj, max_iterations=2, abs_err_tolerance=0.0, rel_err_tolerance=0.0,
                                    linearization_kwargs=dict(kernels=K), linear_solver_kwargs=dict(ordering=ordering))
        solver = opt.linear_solver
        with torch.no_grad():
            info = opt.optimize(damping=1e-2, adaptive_damping=True, ellipsoidal_damping=True, track_err_history=True)
            r = torch.randn(2, solver.linearization.n, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
            w = solver.solve_with_factor(r)
        assert solver.levels == (ordering == "nd")
        if solver.levels:
            t = solver._level_t
            dst = t["blk_dst"].numpy()
            assert ((dst >> 30) & 1).any() and not ((dst >> 30) & 1).all()          # some pairs transposed, not all
            pat = solver.pattern
            assert pat.ntiles == 5 and pat.tree_levels < pat.ntiles
            poc, cop = t["pad_of_col"].numpy(), t["col_of_pad"].numpy()
            assert np.array_equal(cop[poc], np.arange(6 * 96)) and (cop < 0).sum() == pat.npad - 6 * 96
        out[ordering] = (solver.delta.clone(), info.err_history.clone(), w)
    np.testing.assert_allclose(out["nd"][0].numpy(), out["natural"][0].numpy(), rtol=0, atol=1e-10)
    np.testing.assert_allclose(out["nd"][1].numpy(), out["natural"][1].numpy(), rtol=1e-11)
    np.testing.assert_allclose(out["nd"][2].numpy(), out["natural"][2].numpy(), rtol=0, atol=1e-9 * float(out["natural"][2].abs().max()))
