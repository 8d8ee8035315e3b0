"""-m gpu parity tests of the HIP kernels (through the C ABI) against the oracle / golden fixtures."""
import numpy as np
import pytest
import torch

from theseus_amd._lib import THX_ERR_CHUNKS

from oracle import lie as olie
from oracle import pose_graph as opg
from tests.helpers import golden_problem, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from theseus_amd.kernels import default_kernels
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return default_kernels()


def _tol(dtype):
    return 3e-5 if dtype == torch.float32 else 1e-11


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("f64", torch.float64)])
def test_lie_ops_vs_reference_golden(K, tag, dtype):
    """fp64: the reference's outputs to 1e-11.  fp32: the kernels evaluate in fp64 registers
    (csrc/lie.cuh "Evaluation precision"), so they are compared (a) with the exact values -- the fp64
    oracle on the same fp32 inputs with the fp32 thresholds -- at fp32 output rounding, and (b) with the
    reference's fp32 outputs at the reference's OWN distance from those exact values."""
    from tests.helpers import f32_thresholds
    g = load_golden(f"lie_se3_{tag}")
    xi = torch.from_numpy(g["xi"]).cuda()
    X, J = K.se3_exp(xi, jac=True)
    Xg = torch.from_numpy(g["exp"]).cuda()
    Yg = torch.from_numpy(g["Y"]).cuda()
    lg, Jl = K.se3_log(Xg, jac=True)
    got = dict(exp=X, jexp=J, log=lg, jlog=Jl, adj=K.se3_adjoint(Xg), inv=K.se3_inverse(Xg),
               compose=K.se3_compose(Xg, Yg))
    got = {k: v.cpu().numpy() for k, v in got.items()}
    if dtype == torch.float64:
        for k, v in got.items():
            np.testing.assert_allclose(v, g[k], rtol=1e-11, atol=1e-11, err_msg=k)
        return
    with f32_thresholds():
        xi64, X64, Y64 = (torch.from_numpy(g[k]).double() for k in ("xi", "exp", "Y"))
        exact = dict(exp=olie.se3_exp(xi64), adj=olie.se3_adjoint(X64), inv=olie.se3_inverse(X64),
                     compose=olie.se3_compose(X64, Y64))
        exact["log"], exact["jlog"] = olie.se3_log_jlog(X64)
    for k, ex in exact.items():
        ex = ex.numpy()
        rows = lambda a: np.abs(a).reshape(ex.shape[0], -1).max(1)  # noqa: E731
        scale = np.maximum(1.0, rows(ex))
        dev, ref_dev = rows(got[k] - ex), rows(g[k] - ex)
        assert (dev <= 4e-7 * scale).all(), (k, (dev / scale).max())            # (a) fp32 rounding of exact
        assert (rows(got[k] - g[k]) <= ref_dev + 4e-7 * scale).all(), k         # (b) inside the reference's band
    # jexp has no fp64 twin in the oracle module: reference fp32 output, fp32 tolerance
    np.testing.assert_allclose(got["jexp"], g["jexp"], rtol=3e-5, atol=3e-5)


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("f64", torch.float64)])
def test_se2_ops_vs_reference_golden(K, tag, dtype):
    """thx_se2_op against the reference's SE2 outputs (theseus/geometry/se2.py); same criteria as the SE3 test."""
    from oracle import lie_se2
    from tests.helpers import f32_thresholds
    g = load_golden(f"lie_se2_{tag}")
    xi = torch.from_numpy(g["xi"]).cuda()
    X, J = K.se2_exp(xi, jac=True)
    Xg, Yg = torch.from_numpy(g["exp"]).cuda(), torch.from_numpy(g["Y"]).cuda()
    lg, Jl = K.se2_log(Xg, jac=True)
    got = dict(exp=X, jexp=J, log=lg, jlog=Jl, adj=K.se2_adjoint(Xg), inv=K.se2_inverse(Xg), compose=K.se2_compose(Xg, Yg))
    got = {k: v.cpu().numpy() for k, v in got.items()}
    if dtype == torch.float64:
        for k, v in got.items():
            np.testing.assert_allclose(v, g[k], rtol=1e-11, atol=1e-11, err_msg=k)
        return
    with f32_thresholds():
        xi64, X64, Y64 = (torch.from_numpy(g[k]).double() for k in ("xi", "exp", "Y"))
        exact = dict(exp=lie_se2.se2_exp(xi64), adj=lie_se2.se2_adjoint(X64), inv=lie_se2.se2_inverse(X64),
                     compose=lie_se2.se2_compose(X64, Y64))
        exact["log"], exact["jlog"] = lie_se2.se2_log_jlog(X64)
    for k, ex in exact.items():
        ex = ex.numpy()
        rows = lambda a: np.abs(a).reshape(ex.shape[0], -1).max(1)  # noqa: E731
        scale = np.maximum(1.0, rows(ex))
        dev, ref_dev = rows(got[k] - ex), rows(g[k] - ex)
        assert (dev <= 4e-7 * scale).all(), (k, (dev / scale).max())
        assert (rows(got[k] - g[k]) <= ref_dev + 4e-7 * scale).all(), k
    np.testing.assert_allclose(got["jexp"], g["jexp"], rtol=3e-5, atol=3e-5)


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("f64", torch.float64)])
def test_so3_ops_vs_reference_golden(K, tag, dtype):
    """thx_so3_op against the reference's SO3 outputs (torchlie so3_impl.py under theseus/geometry/so3.py), incl. the
    near-zero / d-near-zero / near-pi branches; same criteria as the SE3 test."""
    from oracle import lie_so3
    from tests.helpers import f32_thresholds
    g = load_golden(f"lie_so3_{tag}")
    w = torch.from_numpy(g["xi"]).cuda()
    X, J = K.so3_exp(w, jac=True)
    Xg, Yg = torch.from_numpy(g["exp"]).cuda(), torch.from_numpy(g["Y"]).cuda()
    lg, Jl = K.so3_log(Xg, jac=True)
    got = dict(exp=X, jexp=J, log=lg, jlog=Jl, adj=K.so3_adjoint(Xg), inv=K.so3_inverse(Xg), compose=K.so3_compose(Xg, Yg))
    got = {k: v.cpu().numpy() for k, v in got.items()}
    if dtype == torch.float64:
        for k, v in got.items():
            np.testing.assert_allclose(v, g[k], rtol=1e-11, atol=1e-11, err_msg=k)
        return
    with f32_thresholds():
        w64, X64, Y64 = (torch.from_numpy(g[k]).double() for k in ("xi", "exp", "Y"))
        exact = dict(adj=lie_so3.so3_adjoint(X64), inv=lie_so3.so3_inverse(X64), compose=lie_so3.so3_compose(X64, Y64))
        exact["exp"], exact["jexp"] = lie_so3.so3_exp_jexp(w64)
        exact["log"], exact["jlog"] = lie_so3.so3_log_jlog(X64)
    for k, ex in exact.items():
        ex = ex.numpy()
        rows = lambda a: np.abs(a).reshape(ex.shape[0], -1).max(1)  # noqa: E731
        scale = np.maximum(1.0, rows(ex))
        dev, ref_dev = rows(got[k] - ex), rows(g[k] - ex)
        assert (dev <= 4e-7 * scale).all(), (k, (dev / scale).max())
        assert (rows(got[k] - g[k]) <= ref_dev + 4e-7 * scale).all(), k


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("f64", torch.float64)])
def test_so2_ops_vs_reference_golden(K, tag, dtype):
    """thx_so2_op against the reference's SO2 outputs (theseus/geometry/so2.py:116-117,167-235): the angle <-> [cos, sin] maps,
    the angle-addition compose, the inverse and the unit Jacobians / adjoint; angles through 0, +-pi and beyond.  fp32: within
    fp32 rounding of the exact value, hence inside the reference's own fp32 band (the kernels evaluate in fp64 registers)."""
    from oracle import lie_so2
    g = load_golden(f"lie_so2_{tag}")
    th_ = torch.from_numpy(g["xi"]).cuda()
    X, J = K.so2_exp(th_, jac=True)
    Xg, Yg = torch.from_numpy(g["exp"]).cuda(), torch.from_numpy(g["Y"]).cuda()
    lg, Jl = K.so2_log(Xg, jac=True)
    got = dict(exp=X, jexp=J, log=lg, jlog=Jl, adj=K.so2_adjoint(Xg), inv=K.so2_inverse(Xg), compose=K.so2_compose(Xg, Yg))
    got = {k: v.cpu().numpy() for k, v in got.items()}
    for k in ("jexp", "jlog", "adj", "inv"):
        np.testing.assert_array_equal(got[k], g[k], err_msg=k)          # exact: ones, and a sign flip
    if dtype == torch.float64:
        for k in ("exp", "log", "compose"):
            np.testing.assert_allclose(got[k], g[k], rtol=1e-14, atol=1e-14, err_msg=k)
        return
    th64, X64, Y64 = (torch.from_numpy(g[k]).double() for k in ("xi", "exp", "Y"))
    exact = dict(exp=lie_so2.so2_exp(th64), log=lie_so2.so2_log(X64), compose=lie_so2.so2_compose(X64, Y64))
    for k, ex in exact.items():
        ex = ex.numpy()
        rows = lambda a: np.abs(a).reshape(ex.shape[0], -1).max(1)  # noqa: E731
        scale = np.maximum(1.0, rows(ex))
        dev, ref_dev = rows(got[k] - ex), rows(g[k] - ex)
        assert (dev <= 2e-7 * scale).all(), (k, (dev / scale).max())
        assert (rows(got[k] - g[k]) <= ref_dev + 2e-7 * scale).all(), k


CASES = ["pg_f64_lm", "pg_f32_lm", "pg_f32_lm_b16", "pg_f64_lm_adaptive_ellips", "pg_f64_gn",
         "pg2_f64_lm", "pg2_f32_lm", "pg2_f64_lm_adaptive",   # pg2_*: SE2 (thx_pg2_*)
         "pg3_f64_lm", "pg3_f32_lm", "pg3_f64_lm_adaptive",   # pg3_*: SO3 (thx_pgso3_*)
         "pgso2_f64_lm", "pgso2_f32_lm", "pgso2_f64_lm_adaptive"]   # pgso2_*: SO2 (thx_pgso2_*)


@pytest.mark.parametrize("name", CASES)
def test_assemble_error_jacobians_vs_reference_golden(K, name):
    from tests.gpu_helpers import alloc_dense, sym_from_lower, to_device_problem
    g = load_golden(name)
    p, poses0, kw = golden_problem(g)
    dtype = poses0.dtype
    f32 = dtype == torch.float32
    s, t = to_device_problem(p, poses0)
    ds = s.on("cuda")
    B, n = poses0.shape[0], s.num_cols
    H, gv, ld = alloc_dense(B, n, dtype)
    K.pg_assemble(ds, t, H, gv)
    AtA = sym_from_lower(H, n).cpu().numpy()
    if f32:
        # exact values of what the fp32 reference approximates (tests/helpers.py:f32_truth_problem)
        from tests.helpers import f32_thresholds, f32_truth_problem
        p64, poses64 = f32_truth_problem(p, poses0)
        with f32_thresholds():
            A64, b64 = opg.dense_linearize(p64, poses64)
            H64, g64 = opg.hessian(A64, b64)
            err64 = opg.error_metric(p64, poses64).numpy()
        A64, b64, H64, g64 = (x.numpy() for x in (A64, b64, H64, g64[..., 0]))

        def in_band(ours, ref32, exact, rel):
            """ours within `rel` (fp32 rounding) of exact, hence inside the reference's own fp32 band."""
            sc = np.abs(exact).max()
            dev, ref_dev = np.abs(ours - exact).max(), np.abs(ref32 - exact).max()
            assert dev <= rel * sc, (dev / sc, ref_dev / sc)
            assert np.abs(ours - ref32).max() <= ref_dev + rel * sc
        in_band(AtA, g["AtA"][0], H64, 5e-7)            # blocks accumulated in fp32 from fp64-evaluated J
        in_band(gv.cpu().numpy(), g["Atb"][0][..., 0], g64, 2e-7)   # accumulated in fp64, rounded once
    else:
        # SE2 fp64: the reference's Jlog coefficients half_theta*sine/(1-cosine) and 1/theta - 0.5*sine/(1-cosine)
        # (se2.py:205-214) amplify a 1-ulp difference of the composed cosine (FMA contraction, summation order) by
        # 2/theta^2 resp. 1/(theta (1-cosine)); the fixtures hold a residual rotation of 7.8e-3 rad -> 3e-12 resp.
        # 3e-10 relative: measured here 6e-11 of max|A|, 1.5e-11 of max|AtA|.  Two correct fp64 evaluations of the
        # reference's formula differ by that much, so SE2 is pinned at 1e-9 of scale (SE3: 5e-12, no such term).
        r64 = {"SE2": 1e-9, "SO3": 5e-11}.get(p.group, 5e-12)
        sc = np.abs(g["AtA"][0]).max()
        np.testing.assert_allclose(AtA, g["AtA"][0], rtol=0, atol=sc * r64)
        np.testing.assert_allclose(gv.cpu().numpy(), g["Atb"][0][..., 0], rtol=0, atol=np.abs(g["Atb"][0]).max() * r64)
    # untouched entries stay exactly zero (structure): pattern == block pattern
    pat = np.zeros((n, n), bool)
    d = p.dof
    for r, c in s.lower_block_pattern():
        pat[d * r:d * r + d, d * c:d * c + d] = True
    assert not (H[:, :n, :n].cpu().numpy()[:, ~pat] != 0).any()
    # error metric
    part = torch.empty(THX_ERR_CHUNKS, B, dtype=dtype, device="cuda")
    err = torch.empty(B, dtype=dtype, device="cuda")
    K.pg_error(ds, t, part, err)
    if f32:
        np.testing.assert_allclose(err.cpu().numpy(), err64, rtol=3e-7)
        assert np.abs(err.cpu().numpy() - g["err0"]).max() <= np.abs(g["err0"] - err64).max() + 3e-7 * err64.max()
    else:
        np.testing.assert_allclose(err.cpu().numpy(), g["err0"], rtol=1e-12)
    # Jacobian blocks against the reference's dense A, b
    E, Kp = s.num_edges, s.num_priors
    J0 = torch.empty(E, B, d, d, dtype=dtype, device="cuda"); J1 = torch.empty_like(J0)
    eb = torch.empty(E, B, d, dtype=dtype, device="cuda")
    Jp = torch.empty(Kp, B, d, d, dtype=dtype, device="cuda"); ep = torch.empty(Kp, B, d, dtype=dtype, device="cuda")
    K.pg_jacobians(ds, t, J0, J1, eb, Jp, ep)
    A = np.zeros((B, s.num_rows, n), dtype=g["A0"].dtype); b = np.zeros((B, s.num_rows), dtype=g["A0"].dtype)
    for e in range(E):
        r, i, j = int(s.edge_row_start[e]), int(s.edge_i[e]), int(s.edge_j[e])
        A[:, r:r + d, d * i:d * i + d] = J0[e].cpu().numpy()
        A[:, r:r + d, d * j:d * j + d] = J1[e].cpu().numpy()
        b[:, r:r + d] = -eb[e].cpu().numpy()
    for k in range(Kp):
        r, i = int(s.prior_row_start[k]), int(s.prior_pose[k])
        A[:, r:r + d, d * i:d * i + d] = Jp[k].cpu().numpy()
        b[:, r:r + d] = -ep[k].cpu().numpy()
    if f32:
        in_band(A, g["A0"], A64, 2e-7)
        in_band(b, g["b0"], b64, 2e-7)
    else:
        np.testing.assert_allclose(A, g["A0"], rtol=0, atol=np.abs(g["A0"]).max() * max(r64, 1e-11))
        np.testing.assert_allclose(b, g["b0"], rtol=0, atol=np.abs(g["b0"]).max() * 1e-11 + 1e-30)


def _random_spd(B, n, dtype, seed, cond=1e3):
    gen = torch.Generator().manual_seed(seed)
    A = torch.randn(B, n, n + 8, dtype=torch.float64, generator=gen)
    M = A @ A.transpose(1, 2) / (n + 8)
    M = M + (1.0 / cond) * torch.eye(n, dtype=torch.float64)
    return M.to(dtype)


@pytest.fixture
def split_diag(K):
    """The diagonal phase as chol_syrk_kernel + chol_potrf_kernel at ANY batch size (default: from 2048 problems on)."""
    prev = K.chol_split_diag_min_batch(0)
    yield
    K.chol_split_diag_min_batch(prev)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("n,B", [(6, 3), (48, 5), (126, 4), (128, 9), (132, 3), (258, 8), (390, 17), (1536, 8)])
def test_chol_factor_solve_vs_lapack(K, dtype, n, B, fused):
    _chol_vs_lapack(K, dtype, n, B, fused)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("n,B", [(6, 3), (48, 5), (126, 4), (128, 9), (132, 3), (258, 8), (390, 17), (1536, 8)])
def test_chol_split_diagonal_phase_vs_lapack(K, split_diag, dtype, n, B, fused):
    """The same matrices through the SPLIT diagonal phase: MFMA-only SYRK kernel + the one-wave-per-tile kernel that keeps the
    128 x 128 tile in registers (register x register MFMAs), incl. tiles that straddle the matrix edge (n = 6 ... 390)."""
    _chol_vs_lapack(K, dtype, n, B, fused)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_chol_split_and_fused_diagonal_phase_agree(K, dtype):
    """Both schedules run the same arithmetic on the tile (same MFMA pairings, same 32 x 32 pivot chains): L and the solve panels
    come out BIT-identical; the fused forward substitution sums in a different order (rounding only)."""
    from tests.gpu_helpers import factor_and_solve
    from theseus_amd.kernels import round_up
    n, B = 700, 12
    M = _random_spd(B, n, dtype, seed=77)
    rhs = torch.randn(B, n, dtype=torch.float64, generator=torch.Generator().manual_seed(1)).to(dtype).cuda()
    ld = round_up(n, 32)
    H = torch.zeros(B, ld, ld, dtype=dtype)
    H[:, :n, :n] = torch.tril(M)
    H = H.cuda()
    lam = torch.full((B,), 0.05, dtype=dtype, device="cuda")
    out = {}
    for split in (True, False):
        prev = K.chol_split_diag_min_batch(0 if split else 2 ** 31 - 1)
        try:
            out[split] = factor_and_solve(K, H, n, rhs, damping=lam, ellipsoidal=True, eps=1e-8, fused=True)
        finally:
            K.chol_split_diag_min_batch(prev)
    (La, xa, ia), (Lb, xb, ib) = out[True], out[False]
    assert int(ia.abs().sum()) == 0 and int(ib.abs().sum()) == 0
    assert torch.equal(torch.tril(La[:, :n, :n]), torch.tril(Lb[:, :n, :n]))
    assert (xa - xb).abs().max() <= (2e-5 if dtype == torch.float32 else 1e-13) * xb.abs().max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("ellipsoidal", [False, True])
@pytest.mark.parametrize("n,B", [(384, 3), (640, 8), (1536, 8), (1536, 29), (1024, 32), (600, 5), (1290, 4)])
def test_chol_right_looking_schedule_of_small_batches(K, n, B, ellipsoidal, fused, dtype):
    """fp32, dense frames, <= 32 problems, whole tiles (thx_chol_schedule.right_looking_max_batch): per block column the tile
    factorisation, the substitutions and one workgroup per tile of the trailing matrix.  Against LAPACK in fp64 (L L^T = H + D, the
    solution) at the tolerances of the left-looking tests, and against the left-looking schedule on the same inputs: another
    summation order, the same factor to rounding; the strict upper triangle of L stays zero."""
    from tests.gpu_helpers import factor_and_solve
    f32 = dtype == torch.float32
    M = _random_spd(B, n, dtype, seed=3 * n + B)
    rhs = torch.randn(B, n, dtype=torch.float64, generator=torch.Generator().manual_seed(5)).to(dtype).cuda()
    ld = (n + 127) // 128 * 128                      # whole tiles inside the frame (n = 600 / 1290: the last tile is partial)
    H = torch.zeros(B, ld, ld, dtype=dtype)
    H[:, :n, :n] = torch.tril(M)
    H = H.cuda()
    lam = torch.linspace(0.02, 0.3, B).to(dtype).cuda()
    out = {}
    for rl in (True, False):
        prev = K.chol_right_looking_max_batch(64 if rl else 0)
        try:
            out[rl] = factor_and_solve(K, H, n, rhs, damping=lam, ellipsoidal=ellipsoidal, eps=1e-6, fused=fused)
        finally:
            K.chol_right_looking_max_batch(prev)
    (Lr, xr, ir), (Ll, xl, il) = out[True], out[False]
    assert int(ir.abs().sum()) == 0 and int(il.abs().sum()) == 0
    assert not torch.equal(Lr, Ll)                                        # (it IS another schedule)
    assert float(torch.triu(Lr, 1).abs().max()) == 0.0
    Lr, Ll = Lr[:, :n, :n], Ll[:, :n, :n]
    Md = M.double().cuda()
    dg = torch.diagonal(Md, dim1=1, dim2=2)
    D = lam.double().view(-1, 1) * dg + 1e-6 if ellipsoidal else lam.double().view(-1, 1).expand(B, n)
    Hd = Md + torch.diag_embed(D)
    Lref = torch.linalg.cholesky(Hd)
    scale = Lref.abs().max()
    assert float((Lr.double() - Lref).abs().max() / scale) < (2e-5 if f32 else 1e-13)
    assert float((Lr - Ll).abs().max() / scale) < (2e-5 if f32 else 1e-13)
    xref = torch.cholesky_solve(rhs.double().cpu().unsqueeze(2), Lref.cpu()).squeeze(2).cuda()   # (LAPACK on the host)
    assert float((xr.double() - xref).abs().max() / xref.abs().max()) < (2e-3 if f32 else 1e-10)
    assert float((xr - xl).abs().max() / xref.abs().max()) < (2e-3 if f32 else 1e-10)
    # residual of the fp32 solution in the damped system: at the level of the left-looking schedule's
    res = lambda x: float(((Hd @ x.double().unsqueeze(2)).squeeze(2) - rhs.double()).abs().max() / rhs.abs().max())  # noqa: E731
    assert res(xr) < 2.0 * res(xl) + (1e-6 if f32 else 1e-14)


def test_chol_right_looking_reports_non_positive_definite(K):
    from tests.gpu_helpers import factor_and_solve
    n, B = 640, 6
    M = _random_spd(B, n, torch.float32, seed=9)
    M[2, 300, 300] = -5.0            # breaks positive definiteness in block column 2 of problem 2
    H = torch.tril(M).cuda().contiguous()
    rhs = torch.ones(B, n, dtype=torch.float32, device="cuda")
    prev = K.chol_right_looking_max_batch(64)
    try:
        _, _, info = factor_and_solve(K, H, n, rhs)
    finally:
        K.chol_right_looking_max_batch(prev)
    info = info.cpu()
    assert int(info[2]) != 0 and int(info.ne(0).sum()) == 1


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("n,B", [(258, 5), (390, 12), (700, 9), (1100, 16), (1536, 8)])
def test_chol_column_pairs_are_bit_identical(K, n, B, split):
    """fp32, dense frame: two block columns per off-diagonal launch (chol_offdiag2_f32_kernel: row panel L_i streamed once for
    tiles (i, j) and (i, j + 1), column j's share of the second tile from the first tile's registers) run the same MFMAs in the
    same order as the column-by-column schedule: L, the panels, the fused forward substitution and the solution agree bit for
    bit -- 3 ... 12 tile columns, last tile partial (258, 390, 700, 1100) or full, batch not a multiple of 8."""
    from tests.gpu_helpers import factor_and_solve
    from theseus_amd.kernels import round_up
    dtype = torch.float32
    M = _random_spd(B, n, dtype, seed=n + B)
    rhs = torch.randn(B, n, dtype=torch.float64, generator=torch.Generator().manual_seed(2)).to(dtype).cuda()
    ld = round_up(n, 32)
    H = torch.zeros(B, ld, ld, dtype=dtype)
    H[:, :n, :n] = torch.tril(M)
    H = H.cuda()
    lam = torch.full((B,), 0.05, dtype=dtype, device="cuda")
    out = {}
    prev_split = K.chol_split_diag_min_batch(0 if split else 2 ** 31 - 1)
    try:
        for pairs in (True, False):
            prev = K.chol_column_pairs(pairs)
            try:
                out[pairs] = factor_and_solve(K, H, n, rhs, damping=lam, ellipsoidal=True, eps=1e-8, fused=True)
            finally:
                K.chol_column_pairs(prev)
    finally:
        K.chol_split_diag_min_batch(prev_split)
    (La, xa, ia), (Lb, xb, ib) = out[True], out[False]
    assert int(ia.abs().sum()) == 0 and int(ib.abs().sum()) == 0
    assert torch.equal(torch.tril(La[:, :n, :n]), torch.tril(Lb[:, :n, :n]))
    assert torch.equal(xa, xb)


def test_chol_split_diagonal_phase_reports_non_positive_definite(K, split_diag):
    from tests.gpu_helpers import factor_and_solve
    n, B = 260, 4
    M = _random_spd(B, n, torch.float64, seed=9)
    M[2, 200, 200] = -1.0  # leading minor 201 fails for problem 2 only
    H = torch.zeros(B, 288, 288, dtype=torch.float64); H[:, :n, :n] = torch.tril(M)
    rhs = torch.ones(B, n, dtype=torch.float64)
    _, _, info = factor_and_solve(K, H.cuda(), n, rhs.cuda())
    info = info.cpu().tolist()
    assert info[0] == 0 and info[1] == 0 and info[3] == 0 and info[2] == 201


def _chol_vs_lapack(K, dtype, n, B, fused):
    from tests.gpu_helpers import factor_and_solve
    from theseus_amd.kernels import round_up
    M = _random_spd(B, n, dtype, seed=n + B)
    rhs = torch.randn(B, n, dtype=torch.float64, generator=torch.Generator().manual_seed(1)).to(dtype)
    ld = round_up(n, 32)
    H = torch.zeros(B, ld, ld, dtype=dtype)
    H[:, :n, :n] = torch.tril(M)  # only the lower triangle is meaningful to the solver
    Hd, rd = H.cuda(), rhs.cuda()
    L, x, info = factor_and_solve(K, Hd, n, rd, fused=fused)
    assert int(info.abs().sum()) == 0
    # reference: dense_solver.py:159-161 on the same (fp) matrix, in float64 as the arbiter
    M64 = M.double()
    Lref = torch.linalg.cholesky(M64)
    xref = torch.cholesky_solve(rhs.double().unsqueeze(2), Lref).squeeze(2)
    Lg = torch.tril(L[:, :n, :n]).cpu().double()
    eps = 1.2e-7 if dtype == torch.float32 else 2.3e-16
    # backward error of the factorisation and forward error of the solve, scaled by conditioning
    resid = (Lg @ Lg.transpose(1, 2) - M64).abs().max() / M64.abs().max()
    assert resid < 60 * eps * max(1, n / 64), resid
    xerr = (x.cpu().double() - xref).abs().max() / xref.abs().max()
    assert xerr < (5e-3 if dtype == torch.float32 else 1e-10), xerr
    # H untouched (out-of-place damping contract)
    assert torch.equal(Hd.cpu(), H)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_chol_two_stream_half_batch_schedule(K, dtype):
    """Batches >= THX_CHOL_SPLIT_MIN (default 1024) are factorised as two halves on two streams (chol_kernels.hip:
    factor_impl).  An odd batch of 1029 with per-problem damping and the fused forward substitution: every problem must
    come out as if it had been solved alone -- checked against LAPACK on a sample across both halves."""
    from tests.gpu_helpers import factor_and_solve
    from theseus_amd.kernels import round_up
    n, B = 260, 1029
    gen = torch.Generator(device="cuda").manual_seed(3)
    ld = round_up(n, 32)
    H = torch.zeros(B, ld, ld, dtype=dtype, device="cuda")
    H[:, :n, :n].uniform_(-1, 1, generator=gen)
    H[:, :n, :n] = torch.tril(H[:, :n, :n])
    H.diagonal(dim1=1, dim2=2)[:, :n] += float(n)
    rhs = torch.randn(B, n, dtype=dtype, device="cuda", generator=gen)
    lam = torch.rand(B, dtype=dtype, device="cuda", generator=gen) * 0.1
    L, x, info = factor_and_solve(K, H, n, rhs, damping=lam, ellipsoidal=False, fused=True)
    assert int(info.abs().sum()) == 0
    for b_ in (0, 1, 511, 519, 520, 521, 1027, 1028):   # both sides of the split at 520
        Hl = torch.tril(H[b_, :n, :n]).double().cpu()
        M = Hl + torch.tril(Hl, -1).T + lam[b_].double().cpu() * torch.eye(n, dtype=torch.float64)
        xref = torch.cholesky_solve(rhs[b_].double().cpu().view(-1, 1), torch.linalg.cholesky(M)).view(-1)
        err = (x[b_].double().cpu() - xref).abs().max() / xref.abs().max()
        assert err < (1e-4 if dtype == torch.float32 else 1e-11), (b_, err)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("ellipsoidal", [False, True])
def test_chol_damping_matches_reference_semantics(K, dtype, ellipsoidal):
    from tests.gpu_helpers import factor_and_solve
    n, B = 132, 6
    M = _random_spd(B, n, dtype, seed=3)
    rhs = torch.randn(B, n, dtype=dtype, generator=torch.Generator().manual_seed(5))
    lam = torch.tensor([1e-3, 1e-1, 1.0, 10.0, 0.0, 5e-2], dtype=dtype)
    H = torch.zeros(B, 160, 160, dtype=dtype); H[:, :n, :n] = torch.tril(M)
    _, x, info = factor_and_solve(K, H.cuda(), n, rhs.cuda(), damping=lam.cuda(), ellipsoidal=ellipsoidal, eps=1e-8)
    xref = opg.solve(M.double(), rhs.double().unsqueeze(2), lam.double(), ellipsoidal, 1e-8)
    err = (x.cpu().double() - xref).abs().max() / xref.abs().max()
    assert err < (2e-3 if dtype == torch.float32 else 1e-10), err


def test_chol_reports_non_positive_definite(K):
    from tests.gpu_helpers import factor_and_solve
    n, B = 260, 4
    M = _random_spd(B, n, torch.float64, seed=9)
    M[2, 200, 200] = -1.0  # leading minor 201 fails for problem 2 only
    H = torch.zeros(B, 288, 288, dtype=torch.float64); H[:, :n, :n] = torch.tril(M)
    rhs = torch.ones(B, n, dtype=torch.float64)
    _, _, info = factor_and_solve(K, H.cuda(), n, rhs.cuda())
    info = info.cpu().tolist()
    assert info[0] == 0 and info[1] == 0 and info[3] == 0 and info[2] == 201


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_retract_and_mask(K, dtype):
    gen = torch.Generator().manual_seed(2)
    P, B = 7, 70
    poses = olie.se3_exp(torch.randn(B, P, 6, dtype=dtype, generator=gen))
    delta = 0.3 * torch.randn(B, P * 6, dtype=dtype, generator=gen)
    mask = torch.rand(B, generator=gen) < 0.3
    ref = opg.retract(poses, delta * 0.75, ignore_mask=mask)
    pd = poses.transpose(0, 1).contiguous().cuda()
    out = torch.empty_like(pd)
    K.se3_retract(pd, delta.cuda(), 0.75, mask.to(torch.uint8).cuda(), out)
    got = out.cpu().transpose(0, 1)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=_tol(dtype))
    assert torch.equal(got[mask], poses[mask])  # masked rows are bit identical (torch.where semantics)


@pytest.mark.parametrize("ellipsoidal", [False, True])
def test_lm_accept_matches_reference_formula(K, ellipsoidal):
    gen = torch.Generator().manual_seed(4)
    B, n = 37, 54
    dtype = torch.float64
    delta = torch.randn(B, n, dtype=dtype, generator=gen)
    g = torch.randn(B, n, dtype=dtype, generator=gen)
    H = torch.zeros(B, 64, 64, dtype=dtype)
    dg = torch.rand(B, n, dtype=dtype, generator=gen) + 0.5
    H[:, torch.arange(n), torch.arange(n)] = dg
    lam = 10.0 ** torch.randint(-8, 8, (B,), generator=gen).to(dtype)
    prev = torch.rand(B, dtype=dtype, generator=gen) * 10
    new = prev - torch.randn(B, dtype=dtype, generator=gen)
    damping = lam.view(-1, 1) * (dg if ellipsoidal else 1.0)
    den = (delta * (damping * delta + g)).sum(1) / 2
    rho = (prev - new) / den
    rej = rho <= 0.1
    lam_ref = torch.where(rej, lam * 11.0, lam / 9.0).clamp(1e-7, 1e7)
    lam_d = lam.clone().cuda()
    rej_d = torch.empty(B, dtype=torch.uint8, device="cuda")
    K.lm_accept(delta.cuda(), g.cuda(), H.cuda(), n, lam_d, prev.cuda(), new.cuda(), ellipsoidal, 0.1, 9.0, 11.0, rej_d)
    assert torch.equal(rej_d.cpu().bool(), rej)
    np.testing.assert_allclose(lam_d.cpu().numpy(), lam_ref.numpy(), rtol=1e-14)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_generic_block_assembly_vs_dense_scatter(K, dtype):
    """thx_block_assemble against the reference's recipe: scatter the blocks into dense A, b and form A^T A, A^T b
    (dense_linearization.py:29-62).  Mixed dofs / dims, shared (batch-1) blocks, a variable met by many costs."""
    from theseus_amd.generic import BlockAssembler
    from theseus_amd.kernels import round_up
    gen = torch.Generator().manual_seed(11)
    dofs = [1, 2, 3, 6, 4, 3]
    cols, c0 = [], 0
    for d in dofs:
        cols.append((c0, d)); c0 += d
    n = c0
    cost_vars = [[0, 3], [3], [5, 1, 2], [4, 3], [2, 0], [1], [3, 5], [4, 2, 0, 1]]
    cost_dims = [2, 6, 3, 4, 1, 2, 6, 5]
    B = 70
    Js, es = [], []
    for c, vs in enumerate(cost_vars):
        shared = c in (1, 5)  # batch-1 blocks broadcast over the batch
        Js.append([torch.randn(1 if shared else B, cost_dims[c], dofs[v], dtype=torch.float64, generator=gen).to(dtype).cuda()
                   for v in vs])
        es.append(torch.randn(1 if shared else B, cost_dims[c], dtype=torch.float64, generator=gen).to(dtype).cuda())
    asm = BlockAssembler(cols, cost_vars, cost_dims)
    ld = round_up(n, 32)
    H = torch.zeros(B, ld, ld, dtype=dtype, device="cuda")
    g = torch.zeros(B, n, dtype=dtype, device="cuda")
    asm.assemble(K, Js, es, H, g)
    m = sum(cost_dims)
    A = torch.zeros(B, m, n, dtype=torch.float64)
    b = torch.zeros(B, m, dtype=torch.float64)
    r = 0
    for c, vs in enumerate(cost_vars):
        for s, v in enumerate(vs):
            A[:, r:r + cost_dims[c], cols[v][0]:cols[v][0] + dofs[v]] = Js[c][s].cpu().double()
        b[:, r:r + cost_dims[c]] = -es[c].cpu().double()
        r += cost_dims[c]
    AtA, Atb = A.transpose(1, 2) @ A, (A.transpose(1, 2) @ b.unsqueeze(2)).squeeze(2)
    tol = 1e-6 if dtype == torch.float32 else 1e-13
    Hl = torch.tril(H[:, :n, :n]).cpu().double()
    assert (Hl - torch.tril(AtA)).abs().max() <= tol * AtA.abs().max()
    assert (g.cpu().double() - Atb).abs().max() <= tol * Atb.abs().max()
    # only the block pattern is written
    pat = torch.zeros(n, n, dtype=torch.bool)
    for (r0, c0_, da, db) in asm.lower_block_pattern():
        pat[r0:r0 + da, c0_:c0_ + db] = True
    assert not (H[:, :n, :n].cpu()[:, ~pat] != 0).any()
    # an odd-order system goes through the dense solver (identity padding up to the tile edge)
    from tests.gpu_helpers import factor_and_solve
    Hd = H.clone()
    Hd[:, torch.arange(n), torch.arange(n)] += 1.0
    L, x, info = factor_and_solve(K, Hd, n, g, fused=True)
    assert n % 2 == 1 and int(info.abs().sum()) == 0
    ref = torch.linalg.solve(torch.tril(AtA) + torch.tril(AtA, -1).transpose(1, 2) + torch.eye(n, dtype=torch.float64), Atb)
    assert (x.cpu().double() - ref).abs().max() <= (2e-4 if dtype == torch.float32 else 1e-11) * ref.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype", [((5, 7, 3, 4), torch.float32), ((9, 33, 3), torch.float64), ((1, 2, 4), torch.float32),
                                         ((300, 129, 3), torch.float32)])
def test_copy_where(shape, dtype):
    """thx_copy_where: dst[k, b] <- src[k, b] where mask[b], in place, bit-exact."""
    from theseus_amd.kernels import default_kernels
    K = default_kernels()
    gen = torch.Generator(device="cuda").manual_seed(3)
    src = torch.randn(*shape, dtype=dtype, device="cuda", generator=gen)
    dst = torch.randn(*shape, dtype=dtype, device="cuda", generator=gen)
    mask = torch.rand(shape[1], device="cuda", generator=gen) < 0.4
    want = torch.where(mask.view(1, -1, *([1] * (len(shape) - 2))), src, dst)
    K.copy_where(mask, src, dst)
    assert torch.equal(dst, want)
    dst2 = want.clone()
    K.copy_where(torch.zeros_like(mask), src, dst2)       # nothing selected: untouched
    assert torch.equal(dst2, want)
    K.copy_where(torch.ones_like(mask).view(torch.uint8), src, dst2)  # uint8 mask, everything selected
    assert torch.equal(dst2, src)
