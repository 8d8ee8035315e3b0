"""-m gpu: the BLOCK-COMPACT Hessian (include/theseus_hip.h: thx_hblock_layout) -- thx_pg_assemble_blocks writes the list of
non-zero 6 x 6 blocks, every Cholesky tile gathers its pieces (thx_chol_factor_hblocks).  Same arithmetic as the dense frame:
the assembled values, the factor L, the solve panels and whole LM trajectories must be BIT-identical to the dense-frame path."""
import numpy as np
import pytest
import torch

from tests.helpers import golden_problem, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from theseus_amd.kernels import default_kernels
    return default_kernels()


def _assembled(K, name, order=None):
    from tests.gpu_helpers import alloc_dense, to_device_problem
    g = load_golden(name)
    p, poses0, _ = golden_problem(g)
    s, t = to_device_problem(p, poses0)
    ds = s.on("cuda")
    B, n = poses0.shape[0], s.num_cols
    H, gv, ld = alloc_dense(B, n, poses0.dtype)
    K.pg_assemble(ds, t, H, gv)
    hb = s.hessian_blocks()
    dhb = hb.on("cuda")
    Hc = torch.full((B, hb.bstride), float("nan"), dtype=poses0.dtype, device="cuda")
    g2 = torch.empty_like(gv)
    K.pg_assemble_blocks(ds, t, dhb, Hc, g2)
    return s, hb, dhb, H, gv, Hc, g2, n, ld


@pytest.mark.parametrize("name", ["pg_f64_lm", "pg_f32_lm_b16", "pg_full_f64_lm", "pg_full_f32_lm"])
def test_block_assembly_is_bit_identical_to_the_dense_frame(K, name):
    s, hb, dhb, H, gv, Hc, g2, n, ld = _assembled(K, name)
    assert torch.equal(gv, g2)
    assert not torch.isnan(Hc[:, :hb.elems]).any()            # every block is written in full
    He = torch.zeros_like(H)
    K.hblocks_expand(dhb, Hc, He)
    # the dense kernel also writes the part of a straddling diagonal block that lies ABOVE the tile diagonal (never read by
    # anybody); compare what the factorisation reads: the lower tiles
    tiles = torch.zeros(ld, ld, dtype=torch.bool, device="cuda")
    for ti in range(hb.ntiles):
        for tj in range(ti + 1):
            tiles[128 * ti:128 * ti + 128, 128 * tj:128 * tj + 128] = True
    assert torch.equal(torch.where(tiles, He, torch.zeros_like(He)), torch.where(tiles, H, torch.zeros_like(H)))
    d1 = torch.empty(H.shape[0], n, dtype=H.dtype, device="cuda")
    d2 = torch.empty_like(d1)
    K.diag(H, n, d1)
    K.hblocks_diag(dhb, Hc, d2)
    assert torch.equal(d1, d2)
    # host reference of the layout agrees with the device expansion
    np.testing.assert_array_equal(hb.expand(Hc.cpu().numpy(), ld) * tiles.cpu().numpy(), (He * tiles).cpu().numpy())


# how an off-diagonal tile takes its pieces of H (thx_chol_schedule.hb_scatter_max_pieces): "mfma" -- added by the matrix cores, the
# default for a pose graph's few blocks per tile (tiles with more than the 21 blocks the registers hold take hb_add's overflow chunks:
# tests/test_gpu_ba.py forces a reduced camera system through them); "lds" -- the gather rounds (a bundle adjustment's dense tiles).
# Both bit-identical to the dense frame.
GATHER = {"mfma": -1, "lds": 0}


@pytest.mark.parametrize("gather", list(GATHER))
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("name,ellipsoidal", [("pg_full_f32_lm", False), ("pg_full_f64_lm", True), ("pg_f64_lm", False)])
def test_factor_from_blocks_is_bit_identical(K, name, ellipsoidal, split, gather):
    """thx_chol_factor_hblocks against thx_chol_factor_forward on the dense frame of the same H: (0 - L L^T) + H and H - L L^T
    round identically, so L, the panels, y and info agree bit for bit -- fused and split diagonal phase, 12-tile and 1-tile n,
    pieces added by the matrix cores (hb_scatter) and gathered through LDS."""
    s, hb, dhb, H, gv, Hc, g2, n, ld = _assembled(K, name)
    B = H.shape[0]
    nt = (n + 127) // 128
    if nt > 1:   # the default really is the matrix-core path here
        assert 1 <= dhb.c.max_tile_pieces <= 64
    lam = torch.full((B,), 1e-3, dtype=H.dtype, device="cuda")
    prev = K.chol_split_diag_min_batch(0 if split else 2 ** 31 - 1)
    prev_g = K.chol_hb_scatter_max_pieces(GATHER[gather])
    try:
        out = []
        for compact in (False, True):
            L = torch.zeros_like(H)
            panels = torch.zeros(B, nt, 128, 128, dtype=H.dtype, device="cuda")
            info = torch.empty(B, dtype=torch.int32, device="cuda")
            y = torch.empty_like(gv)
            if compact:
                K.chol_factor_hblocks(dhb, Hc, n, lam, ellipsoidal, 1e-8, L, panels, info, rhs=gv, y=y)
            else:
                K.chol_factor(H, n, lam, ellipsoidal, 1e-8, L, panels, info, rhs=gv, y=y)
            out.append((torch.tril(L[:, :n, :n]), panels, y, info))
    finally:
        K.chol_split_diag_min_batch(prev)
        K.chol_hb_scatter_max_pieces(prev_g)
    (La, Pa, ya, ia), (Lb, Pb, yb, ib) = out
    assert int(ia.abs().sum()) == 0 and int(ib.abs().sum()) == 0
    assert torch.equal(La, Lb) and torch.equal(ya, yb)
    for u in range(4):          # the ten lower sub-blocks of every panel (the others are never written)
        for v in range(u + 1):
            assert torch.equal(Pa[:, :, 32 * u:32 * u + 32, 32 * v:32 * v + 32], Pb[:, :, 32 * u:32 * u + 32, 32 * v:32 * v + 32])


@pytest.mark.parametrize("gather", list(GATHER))
@pytest.mark.parametrize("split", [False, True])
def test_factor_from_blocks_in_column_pairs_is_bit_identical(K, split, gather):
    """thx_chol_factor_hblocks, fp32, 12 tile columns: the column-pair schedule (both H tiles of a workgroup taken from the
    block list) against the column-by-column one -- L, panels, y bit for bit."""
    s, hb, dhb, H, gv, Hc, g2, n, ld = _assembled(K, "pg_full_f32_lm")
    B = H.shape[0]
    nt = (n + 127) // 128
    lam = torch.full((B,), 1e-3, dtype=H.dtype, device="cuda")
    prev_split = K.chol_split_diag_min_batch(0 if split else 2 ** 31 - 1)
    prev_g = K.chol_hb_scatter_max_pieces(GATHER[gather])
    out = []
    try:
        for pairs in (True, False):
            prev = K.chol_column_pairs(pairs)
            try:
                L = torch.zeros_like(H)
                panels = torch.zeros(B, nt, 128, 128, dtype=H.dtype, device="cuda")
                info = torch.empty(B, dtype=torch.int32, device="cuda")
                y = torch.empty_like(gv)
                K.chol_factor_hblocks(dhb, Hc, n, lam, False, 1e-8, L, panels, info, rhs=gv, y=y)
                out.append((torch.tril(L[:, :n, :n]), panels, y, info))
            finally:
                K.chol_column_pairs(prev)
    finally:
        K.chol_split_diag_min_batch(prev_split)
        K.chol_hb_scatter_max_pieces(prev_g)
    (La, Pa, ya, ia), (Lb, Pb, yb, ib) = out
    assert int(ia.abs().sum()) == 0 and int(ib.abs().sum()) == 0
    assert torch.equal(La, Lb) and torch.equal(ya, yb)
    for u in range(4):
        for v in range(u + 1):
            assert torch.equal(Pa[:, :, 32 * u:32 * u + 32, 32 * v:32 * v + 32], Pb[:, :, 32 * u:32 * u + 32, 32 * v:32 * v + 32])


@pytest.mark.parametrize("solver", ["dense", "sparse"])
@pytest.mark.parametrize("name", ["pg_full_f32_lm", "pg_full_f64_lm"])
def test_lm_with_block_hessian_equals_dense_frame(name, solver):
    """Whole LM runs (256 poses / 1024 edges, adaptive ellipsoidal damping so that diagonal_scaling / the rho test read the
    block list too), dense solver and tile-sparse solver under its RCM ordering: block-compact == dense frame, bit for bit."""
    import theseus_amd as th
    from tests.test_gpu_lm import build_objective
    g = load_golden(name)
    res = []
    for compact in (True, False):
        obj, _ = build_objective(th, g)
        cls = th.HipCholeskySolver if solver == "dense" else th.HipSparseCholeskySolver
        opt = th.LevenbergMarquardt(obj, linear_solver_cls=cls, max_iterations=4, abs_err_tolerance=0.0, rel_err_tolerance=0.0,
                                    linearization_kwargs=dict(block_hessian=compact),
                                    linear_solver_kwargs=dict(ordering="rcm") if solver == "sparse" else None)
        lin = opt.linear_solver.linearization
        assert lin._compact == compact
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(
            damping=1e-3, adaptive_damping=True, ellipsoidal_damping=True, track_err_history=True))
        res.append((torch.stack([sol[f"pose_{k}"] for k in range(int(g["P"]))], 1), info.err_history, lin))
    (xa, ha, lina), (xb, hb_, linb) = res
    assert torch.equal(xa, xb) and torch.equal(ha, hb_)
    assert lina._H is None                      # the dense frame was never materialised ...
    AtA = lina.AtA                              # ... until somebody reads AtA (thx_hblocks_expand)
    assert torch.equal(AtA, linb.AtA)


@pytest.mark.parametrize("compact", [True, False])
def test_fp64_eight_wave_and_half_tile_offdiag_kernels_are_bit_identical(K, compact):
    """thx_chol_schedule.f64_wide_max_ktiles / f64_half_max_ktiles: the off-diagonal tiles of the first block columns from eight-wave
    workgroups (16 rows of the tile per wave) or as two half tiles from four-wave workgroups (four per CU) -- the same MFMAs in the
    same order as the four-wave kernel: L, y bit for bit, for the block-compact H (matrix-core scatter) and the dense frame, every
    setting from "no column" to "all of them", and the defaults."""
    s, hb, dhb, H, gv, Hc, g2, n, ld = _assembled(K, "pg_full_f64_lm")
    B = H.shape[0]
    nt = (n + 127) // 128
    lam = torch.full((B,), 1e-3, dtype=H.dtype, device="cuda")
    out = []
    for wide, half in ((0, 0), (1, 0), (3, 0), (nt, 0), (0, 1), (0, 5), (0, nt), (nt, 4), (-1, -1)):
        prev = K.chol_f64_wide_max_ktiles(wide)
        prev_h = K.chol_f64_half_max_ktiles(half)
        try:
            L = torch.zeros_like(H)
            panels = torch.zeros(B, nt, 128, 128, dtype=H.dtype, device="cuda")
            info = torch.empty(B, dtype=torch.int32, device="cuda")
            y = torch.empty_like(gv)
            if compact:
                K.chol_factor_hblocks(dhb, Hc, n, lam, True, 1e-8, L, panels, info, rhs=gv, y=y)
            else:
                K.chol_factor(H, n, lam, True, 1e-8, L, panels, info, rhs=gv, y=y)
            out.append((torch.tril(L[:, :n, :n]), y, info))
        finally:
            K.chol_f64_wide_max_ktiles(prev)
            K.chol_f64_half_max_ktiles(prev_h)
    for La, ya, ia in out:
        assert int(ia.abs().sum()) == 0
        assert torch.equal(La, out[0][0]) and torch.equal(ya, out[0][1])
