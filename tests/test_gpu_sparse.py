"""-m gpu: thx_chol_factor_sparse (tile-sparse Cholesky, BaspachoSparseSolver's role) on the HIP kernels: bit-identical to the
dense factorisation of the same banded matrices (skipped tiles are exact zeros), and LM on a large chain-like pose graph
with HipSparseCholeskySolver == the dense solver."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _banded_spd(B, n, bw_tiles, dtype, seed):
    """block-banded SPD matrices: 6x6 blocks within ``bw_tiles`` tiles of the diagonal (+ a few far blocks -> fill)."""
    from theseus_amd.sparse import TilePattern
    P = n // 6
    rng = np.random.default_rng(seed)
    reach = bw_tiles * 21
    blocks = {(p, p) for p in range(P)} | {(p, p - 1) for p in range(1, P)}
    blocks |= {(p, max(0, p - int(rng.integers(2, reach)))) for p in range(2, P, 3)}
    blocks |= {(P - 5, 10)}                                    # one long-range block: fill along the last block rows
    blocks = np.array(sorted(blocks))
    gen = torch.Generator().manual_seed(seed)
    H = torch.zeros(B, n, n, dtype=torch.float64)
    for r, c in blocks:
        H[:, 6 * r:6 * r + 6, 6 * c:6 * c + 6] = torch.randn(B, 6, 6, dtype=torch.float64, generator=gen)
    H = torch.tril(H) + torch.tril(H, -1).transpose(1, 2) + 60.0 * torch.eye(n, dtype=torch.float64)
    return H.to(dtype), TilePattern(n, blocks, 6)


@pytest.mark.parametrize("split_diag", [False, True])
@pytest.mark.parametrize("dtype,n,B", [(torch.float32, 3072, 24), (torch.float64, 1536, 8), (torch.float32, 1530, 1100)])
def test_sparse_factorisation_is_bit_identical_to_dense(dtype, n, B, split_diag):
    from theseus_amd.kernels import default_kernels
    prev = default_kernels().chol_split_diag_min_batch(0 if split_diag else 2 ** 31 - 1)
    prev_rl = default_kernels().chol_right_looking_max_batch(0)   # (the dense side left-looking too, whatever its batch size)
    try:
        _sparse_vs_dense(dtype, n, B)
    finally:
        default_kernels().chol_split_diag_min_batch(prev)
        default_kernels().chol_right_looking_max_batch(prev_rl)


def _sparse_vs_dense(dtype, n, B):
    from theseus_amd.kernels import default_kernels, round_up
    K = default_kernels()
    Hc, pat = _banded_spd(B, n, 2, dtype, seed=n)
    assert pat.l_tiles < pat.ntiles * (pat.ntiles + 1) // 2
    ld = round_up(n, 32)
    H = torch.zeros(B, ld, ld, dtype=dtype, device="cuda")
    H[:, :n, :n] = Hc.cuda()
    lam = torch.full((B,), 0.5, dtype=dtype, device="cuda")
    rhs = torch.randn(B, n, dtype=dtype, device="cuda")
    nt = (n + 127) // 128
    out = []
    for sparse in (False, True):
        L = torch.zeros_like(H)
        panels = torch.zeros(B, nt, 128, 128, dtype=dtype, device="cuda")
        info = torch.empty(B, dtype=torch.int32, device="cuda")
        y, x = torch.empty_like(rhs), torch.empty_like(rhs)
        if sparse:
            K.chol_factor_sparse(H, n, lam, True, 1e-8, L, panels, info, pat, rhs=rhs, y=y)
        else:
            K.chol_factor(H, n, lam, True, 1e-8, L, panels, info, rhs=rhs, y=y)
        x2 = torch.empty_like(rhs)
        if sparse:   # the list-driven solves (thx_chol_solve_sparse): second half after the fused forward, and both halves
            K.chol_solve_sparse(L, n, panels, y, x, pat, backward_only=True)
            K.chol_solve_sparse(L, n, panels, rhs, x2, pat)
        else:
            K.chol_solve_backward(L, n, panels, y, x)
            K.chol_solve(L, n, panels, rhs, x2)
        assert int(info.abs().sum()) == 0
        out.append((torch.tril(L[:, :n, :n]).clone(), y.clone(), x.clone(), x2.clone()))
    (Ld, yd, xd, x2d), (Ls, ys, xs, x2s) = out
    assert torch.equal(Ld, Ls) and torch.equal(yd, ys) and torch.equal(xd, xs) and torch.equal(x2d, x2s)
    x3 = rhs.clone()          # in place
    K.chol_solve_sparse(L, n, panels, x3, x3, pat)
    assert torch.equal(x3, x2s)
    # and it IS the factor: residual against fp64
    Hd = Hc[:2].double().cuda()
    Hd = Hd + torch.diag_embed(0.5 * Hd.diagonal(dim1=1, dim2=2) + 1e-8)
    Lc = Ls[:2].double()
    assert ((Lc @ Lc.transpose(1, 2) - Hd).abs().max() / Hd.abs().max()).item() < (5e-6 if dtype == torch.float32 else 1e-13)


@pytest.mark.parametrize("ordering", ["rcm", "nd"])
def test_lm_on_a_large_chain_graph_sparse_equals_dense(ordering):
    """560 SE3 poses (n = 3360, 27 tiles; fp64 -- the DENSE solver's triangular-solve kernels keep the right-hand side in LDS,
    which bounds fp64 at n <= 3680, fp32 at n <= ~23000), shuffled labels: the sparse solver (RCM ordering + tile pattern)
    reproduces the dense solver's LM run; the pattern prunes most of the tile products."""
    import theseus_amd as th
    from tests.test_sparse_solver import chain_graph
    P, B, dtype = 560, 4, torch.float64
    edges = chain_graph(P, stride=7, span=5, seed=2)
    K = th.default_kernels()
    gen = torch.Generator(device="cuda").manual_seed(7)
    rnd = lambda nn, s: K.se3_exp(s * (2 * torch.rand(nn, 6, dtype=dtype, device="cuda", generator=gen) - 1))  # noqa: E731
    gt = rnd(B * P, 1.5).view(B, P, 3, 4)
    poses0 = K.se3_compose(gt.reshape(-1, 3, 4), rnd(B * P, 0.05)).view(B, P, 3, 4)
    meas = [K.se3_compose(K.se3_compose(K.se3_inverse(gt[:, i].contiguous()), gt[:, j].contiguous()), rnd(B, 0.01)) for (i, j) in edges]

    def run(solver_cls):
        obj = th.Objective(dtype=dtype)
        pv = [th.SE3(tensor=poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
        w = th.ScaleCostWeight(torch.tensor(5.0, dtype=dtype, device="cuda"))
        for k, (i, j) in enumerate(edges):
            obj.add(th.Between(pv[i], pv[j], th.SE3(tensor=meas[k].clone(), name=f"m_{k}"), w, name=f"b_{k}"))
        obj.add(th.Difference(pv[edges[0][0]], th.SE3(tensor=gt[:, edges[0][0]].clone(), name="anchor"), w, name="prior"))
        opt = th.LevenbergMarquardt(obj, linear_solver_cls=solver_cls, max_iterations=5, abs_err_tolerance=0.0, rel_err_tolerance=0.0,
                                    linear_solver_kwargs=dict(ordering=ordering) if solver_cls is th.HipSparseCholeskySolver else None)
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(damping=1e-2, track_err_history=True))
        return torch.stack([sol[f"pose_{k}"] for k in range(P)], 1), info, opt
    dense, dinfo, _ = run(th.HipCholeskySolver)
    sparse, sinfo, opt = run(th.HipSparseCholeskySolver)
    pat = opt.linear_solver.pattern
    assert opt.linear_solver.levels == (ordering == "nd")
    if ordering == "nd":   # elimination-tree parallelism: a handful of dependent launch levels instead of one per block column
        assert pat.tree_levels <= 6 < pat.ntiles
    print(f"[sparse LM] tiles of L: {pat.l_tiles} of {pat.ntiles * (pat.ntiles + 1) // 2}; tile products {pat.tile_products} vs dense {pat.dense_tile_products}")
    assert pat.tile_products * 5 < pat.dense_tile_products
    np.testing.assert_allclose(sparse.cpu().numpy(), dense.cpu().numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(sinfo.err_history.numpy(), dinfo.err_history.numpy(), rtol=1e-9)
    assert sinfo.err_history[:, -1].mean() < 0.05 * sinfo.err_history[:, 0].mean()


def test_sparse_solver_has_no_size_limit_in_fp64():
    """1000 SE3 poses in fp64 (n = 6000): beyond the LDS plan of the dense-frame triangular solves (n <= 3680 in fp64), which the
    list-driven solves do not have.  LM converges; the linear solve of the last iteration satisfies H delta = g."""
    import theseus_amd as th
    from tests.test_sparse_solver import chain_graph
    P, B, dtype = 1000, 2, torch.float64
    edges = chain_graph(P, stride=7, span=5, seed=3)
    K = th.default_kernels()
    gen = torch.Generator(device="cuda").manual_seed(11)
    rnd = lambda nn, s: K.se3_exp(s * (2 * torch.rand(nn, 6, dtype=dtype, device="cuda", generator=gen) - 1))  # noqa: E731
    gt = rnd(B * P, 1.5).view(B, P, 3, 4)
    poses0 = K.se3_compose(gt.reshape(-1, 3, 4), rnd(B * P, 0.05)).view(B, P, 3, 4)
    meas = [K.se3_compose(K.se3_compose(K.se3_inverse(gt[:, i].contiguous()), gt[:, j].contiguous()), rnd(B, 0.01)) for (i, j) in edges]
    obj = th.Objective(dtype=dtype)
    pv = [th.SE3(tensor=poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
    w = th.ScaleCostWeight(torch.tensor(5.0, dtype=dtype, device="cuda"))
    for k, (i, j) in enumerate(edges):
        obj.add(th.Between(pv[i], pv[j], th.SE3(tensor=meas[k].clone(), name=f"m_{k}"), w, name=f"b_{k}"))
    obj.add(th.Difference(pv[edges[0][0]], th.SE3(tensor=gt[:, edges[0][0]].clone(), name="anchor"), w, name="prior"))
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipSparseCholeskySolver, max_iterations=4, abs_err_tolerance=0.0,
                                rel_err_tolerance=0.0)
    sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(damping=1e-2, track_err_history=True))
    assert info.err_history[:, -1].mean() < 0.05 * info.err_history[:, 0].mean()
    solver, lin = opt.linear_solver, opt.linear_solver.linearization
    lin.linearize()
    delta = solver.solve(damping=1e-2, ellipsoidal_damping=False)
    Hf = lin.AtA + 1e-2 * torch.eye(lin.n, dtype=dtype, device="cuda")
    r = (Hf @ delta.unsqueeze(2)).squeeze(2) - lin.Atb.squeeze(2)
    assert (r.abs().max() / lin.Atb.abs().max()).item() < 1e-10
    x = solver.solve_with_factor(lin.Atb.squeeze(2).contiguous())       # both halves with the cached factor
    np.testing.assert_allclose(x.cpu().numpy(), delta.cpu().numpy(), rtol=0, atol=1e-9 * float(delta.abs().max()))


def _chain_problem(th, P, B, dtype, seed=7):
    from tests.test_sparse_solver import chain_graph
    edges = chain_graph(P, stride=7, span=5, seed=2)
    K = th.default_kernels()
    gen = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda nn, s: K.se3_exp(s * (2 * torch.rand(nn, 6, dtype=dtype, device="cuda", generator=gen) - 1))  # noqa: E731
    gt = rnd(B * P, 1.5).view(B, P, 3, 4)
    poses0 = K.se3_compose(gt.reshape(-1, 3, 4), rnd(B * P, 0.05)).view(B, P, 3, 4)
    meas = [K.se3_compose(K.se3_compose(K.se3_inverse(gt[:, i].contiguous()), gt[:, j].contiguous()), rnd(B, 0.01)) for (i, j) in edges]

    def build(**solver_kw):
        obj = th.Objective(dtype=dtype)
        pv = [th.SE3(tensor=poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
        w = th.ScaleCostWeight(torch.tensor(5.0, dtype=dtype, device="cuda"))
        for k, (i, j) in enumerate(edges):
            obj.add(th.Between(pv[i], pv[j], th.SE3(tensor=meas[k].clone(), name=f"m_{k}"), w, name=f"b_{k}"))
        obj.add(th.Difference(pv[edges[0][0]], th.SE3(tensor=gt[:, edges[0][0]].clone(), name="anchor"), w, name="prior"))
        return th.LevenbergMarquardt(obj, linear_solver_cls=th.HipSparseCholeskySolver, max_iterations=4, abs_err_tolerance=0.0,
                                     rel_err_tolerance=0.0, linear_solver_kwargs=solver_kw)
    return build


@pytest.mark.parametrize("dtype,P", [(torch.float32, 700), (torch.float64, 300)])
def test_tile_packed_factor_is_bit_identical_to_the_dense_frame(dtype, P):
    """HipSparseCholeskySolver keeps L TILE-PACKED by default -- (B, nslots, 128, 128): only the tiles of the pattern exist -- when
    the Hessian is block-compact.  Same kernels, same arithmetic: the LM run, the factor (unpacked) and a cached-factor solve are
    bit-identical to the dense-frame factor of the same pattern."""
    import theseus_amd as th
    B = 6
    build = _chain_problem(th, P, B, dtype)
    out = {}
    for packed in (True, False):
        opt = build(packed_factor=packed, ordering="rcm")     # (the column-by-column schedule: same arithmetic in both layouts)
        solver = opt.linear_solver
        assert solver.packed_factor == packed
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(damping=1e-2, adaptive_damping=True, track_err_history=True))
        x = solver.solve_with_factor(solver.linearization.g.clone())
        out[packed] = (torch.stack([sol[f"pose_{k}"] for k in range(P)], 1), info.err_history, solver.dense_factor().clone(), x, solver)
    (xa, ha, La, sa, solver), (xb, hb, Lb, sb, dsolver) = out[True], out[False]
    pat = solver.pattern
    assert solver.L.dim() == 4 and solver.L.shape[1] == pat.nslots == pat.ntiles + len(pat.tables["col_row"])
    n = solver.linearization.n
    assert torch.equal(torch.tril(La[:, :n, :n]), torch.tril(Lb[:, :n, :n]))
    assert torch.equal(xa, xb) and torch.equal(ha, hb) and torch.equal(sa, sb)
    assert ha[:, -1].mean() < 0.05 * ha[:, 0].mean()
    print(f"[packed factor] n = {n}: {pat.nslots} tiles of 128 x 128 = {pat.nslots * 65536 * xa.element_size() / 4 / 1e6:.1f} MB per problem "
          f"against {dsolver.L[0].numel() * xa.element_size() / 1e6:.1f} MB of dense frame")


def test_tile_packed_factor_runs_a_4096_pose_graph_at_the_reference_sweeps_batch_size():
    """evaluations/pose_graph_synthetic.sh sweeps to 4096 poses at batch sizes up to 256: n = 24576 -- a dense (B, ld, ld) frame
    of L would be 2.4 GB per problem in fp32 (batch 256: 618 GB, twice the HBM of an MI355X; H another 618 GB before round 3).
    Block-compact H + tile-packed L: a few tens of MB per problem."""
    import theseus_amd as th
    P, B, dtype = 4096, 256, torch.float32
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    build = _chain_problem(th, P, B, dtype)
    opt = build()
    solver = opt.linear_solver
    with torch.no_grad():
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(damping=1e-2, track_err_history=True))
    assert solver.packed_factor and solver.linearization._compact and solver.linearization._H is None
    peak = torch.cuda.max_memory_allocated() / 1e9
    pat = solver.pattern
    print(f"[4096 poses, batch 256] L: {pat.nslots} tiles = {solver.L[0].numel() * 4 / 1e6:.1f} MB per problem (dense frame: "
          f"{(6 * P) ** 2 * 4 / 1e6:.0f} MB); peak device memory of the run {peak:.1f} GB; cost {info.err_history[:, 0].mean():.1f} -> "
          f"{info.err_history[:, -1].mean():.4f}")
    assert int(solver.info.abs().sum()) == 0
    assert info.err_history[:, -1].mean() < 0.05 * info.err_history[:, 0].mean()
    assert peak < 60.0


def test_fp64_beyond_the_fused_forward_substitution_limit():
    """1700 SE3 poses in fp64: n = 10200 -- more than the fused forward substitution can keep in LDS next to the diagonal tile
    (~9 k in fp64).  The solver then factorises without it and runs both list-driven triangular solves: the run converges and
    the last linear system is solved to rounding."""
    import theseus_amd as th
    P, B, dtype = 1700, 3, torch.float64
    opt = _chain_problem(th, P, B, dtype)(ordering="rcm")   # (the fused forward substitution belongs to the column-by-column schedule)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # (a refused factorisation would surface as the loop's RuntimeWarning)
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(damping=1e-2, track_err_history=True))
    solver, lin = opt.linear_solver, opt.linear_solver.linearization
    assert solver._unfused_forward and torch.isfinite(info.err_history).all()
    assert info.err_history[:, -1].mean() < 0.05 * info.err_history[:, 0].mean()
    lin.linearize()
    delta = solver.solve(damping=1e-2, ellipsoidal_damping=False)
    Hf = lin.AtA + 1e-2 * torch.eye(lin.n, dtype=dtype, device="cuda")
    r = (Hf @ delta.unsqueeze(2)).squeeze(2) - lin.Atb.squeeze(2)
    assert (r.abs().max() / lin.Atb.abs().max()).item() < 1e-10


# ---- the level-scheduled solver (tile-level nested dissection, thx_chol_factor_levels / thx_chol_solve_levels) -----------------
@pytest.mark.parametrize("dtype,P,B", [(torch.float32, 700, 6), (torch.float64, 300, 3), (torch.float32, 1500, 40)])
def test_level_schedule_factor_and_solves(dtype, P, B):
    """The factor the level schedule leaves in the tile-packed buffer IS the Cholesky factor of the damped Hessian in the padded
    order (identity on the padding), the solves invert it, a snapshot of it solves the same, and the split / fused diagonal
    schedules (picked per level by its workgroup count) give the same bits."""
    import theseus_amd as th
    opt = _chain_problem(th, P, B, dtype)(ordering="nd")
    solver, lin = opt.linear_solver, opt.linear_solver.linearization
    pat = solver.pattern
    assert solver.levels and solver.packed_factor and lin._compact and pat.tree_levels < pat.ntiles
    opt.objective.update()
    lin.linearize()
    out = {}
    for split_min in (1, 1 << 30):          # every level through chol_syrk + chol_potrf / through the fused chol_diag
        prev = th.default_kernels().chol_split_diag_min_batch(split_min)
        try:
            delta = solver.solve(damping=0.05, ellipsoidal_damping=True, damping_eps=1e-8)
        finally:
            th.default_kernels().chol_split_diag_min_batch(prev)
        out[split_min] = (delta.clone(), solver.L.clone())
    assert torch.equal(out[1][0], out[1 << 30][0]) and torch.equal(out[1][1], out[1 << 30][1])
    assert int(solver.info.abs().sum()) == 0
    n, npad = lin.n, pat.npad
    H = lin.AtA.double()
    Hd = H + torch.diag_embed(0.05 * H.diagonal(dim1=1, dim2=2) + 1e-8)
    idx = torch.from_numpy(pat.pad_of_col).long().cuda()
    Lp = solver.dense_factor()[:2].double()
    assert Lp.shape[-1] == npad
    LLt = Lp @ Lp.transpose(1, 2)
    tol = 5e-6 if dtype == torch.float32 else 1e-13
    assert ((LLt[:, idx[:, None], idx[None, :]] - Hd[:2]).abs().max() / Hd.abs().max()).item() < tol
    pad = torch.ones(npad, dtype=torch.bool, device="cuda")
    pad[idx] = False
    assert float(Lp[:, pad, :].abs().max()) == 0.0 and float(Lp[:, :, pad].abs().max()) == 0.0   # padding: never written
    g = lin.Atb.squeeze(2)
    r = (Hd @ delta.double().unsqueeze(2)).squeeze(2) - g.double()
    assert (r.abs().max() / g.abs().max()).item() < (2e-2 if dtype == torch.float32 else 1e-10)
    # the cached-factor solve (both halves through thx_chol_solve_levels) against the step whose forward half was fused into the
    # factorisation: same factor, another summation order in the forward substitution
    x = solver.solve_with_factor(g.contiguous())
    stol = (2e-4 if dtype == torch.float32 else 1e-11) * float(delta.abs().max())
    assert float((x - delta).abs().max()) <= stol and torch.isfinite(delta).all()
    snap = solver.factor_snapshot()
    solver.solve(damping=7.0, ellipsoidal_damping=False)            # the solver factorises something else ...
    assert torch.equal(solver.solve_with_snapshot(snap, g.contiguous()), x)   # ... the snapshot still solves the first system


def test_level_schedule_not_positive_definite_is_reported():
    import theseus_amd as th
    opt = _chain_problem(th, 400, 3, torch.float32)(ordering="nd")
    solver, lin = opt.linear_solver, opt.linear_solver.linearization
    opt.objective.update()
    lin.linearize()
    with pytest.raises(RuntimeError, match="not positive-definite"):
        solver.solve(damping=-1e6, ellipsoidal_damping=False)


def test_full_size_implicit_gradients_through_the_level_schedule():
    """configs[4]'s shape (256 poses / 1024 edges with RANDOM loop closures: separators are wide, the elimination tree is shallow
    but not a chain) through HipSparseCholeskySolver(ordering="nd"): forward LM, the undamped last step, the backward solve with
    the cached level-scheduled factor -- against the gradients the REAL reference produced (tests/golden/pg_full_f64_implicit)."""
    import theseus_amd as th
    from tests.helpers import load_golden
    from tests.implicit_common import check_full_size_implicit, run_implicit
    g = load_golden("pg_full_f64_implicit")
    final, loss, grads, info, opt, _ = run_implicit(th, g, "cuda", gauge_free=True,
                                                    solver=dict(linear_solver_cls=th.HipSparseCholeskySolver,
                                                                linear_solver_kwargs=dict(ordering="nd")))
    assert opt.linear_solver.levels and opt.linear_solver.pattern.tree_levels < opt.linear_solver.pattern.ntiles
    check_full_size_implicit(g, final, loss, grads, "full size implicit fp64, level schedule")
