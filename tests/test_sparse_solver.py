"""Tile-sparse Cholesky solver (theseus_amd/sparse.py): symbolic tile pattern, fill-reducing variable ordering, and the host
path end to end on a chain-like pose graph (TEST stand-in kernels here; the HIP kernels in tests/test_gpu_sparse.py)."""
import numpy as np
import pytest
import torch


def chain_graph(P, stride=7, span=5, seed=0, shuffle=True):
    """odometry chain + local loop closures; pose NAMES are shuffled so that insertion order is far from banded."""
    from theseus_amd.utils.synthetic import chain_graph_topology
    return chain_graph_topology(P, stride=stride, span=span, seed=seed, shuffle=shuffle)


def test_tile_pattern_covers_the_numeric_fill():
    from theseus_amd.sparse import TilePattern
    rng = np.random.default_rng(3)
    P, dof = 120, 6
    n = P * dof
    blocks = {(p, p) for p in range(P)} | {(p, p - 1) for p in range(1, P)} | {(p, p - 9) for p in range(9, P, 4)} | {(77, 3), (110, 40)}
    blocks = np.array(sorted(blocks))
    tp = TilePattern(n, blocks, dof)
    A = np.zeros((n, n))
    for r, c in blocks:
        A[dof * r:dof * r + dof, dof * c:dof * c + dof] = rng.standard_normal((dof, dof))
    A = np.tril(A) + np.tril(A, -1).T + 40 * np.eye(n)
    L = np.linalg.cholesky(A)
    nt = tp.ntiles
    Lp = np.zeros((nt * 128, nt * 128))
    Lp[:n, :n] = L
    tiles = np.abs(Lp.reshape(nt, 128, nt, 128)).max(axis=(1, 3)) > 0
    assert not (tiles & ~tp.lower).any()                       # the symbolic pattern covers every numeric non-zero tile
    assert tp.l_tiles < nt * (nt + 1) // 2                     # ... and is sparser than the dense lower triangle
    t = tp.tables
    for j in range(nt):                                        # tables: CSR over columns, K-lists = row-pattern intersections
        rows = t["col_row"][t["col_ptr"][j]:t["col_ptr"][j + 1]].tolist()
        assert sorted(rows) == (np.nonzero(tp.lower[j + 1:, j])[0] + j + 1).tolist() and tp.col_count[j] == len(rows)
        assert t["diag_k"][t["diag_kptr"][j]:t["diag_kptr"][j + 1]].tolist() == np.nonzero(tp.lower[j, :j])[0].tolist()
        for e, i in zip(range(t["col_ptr"][j], t["col_ptr"][j + 1]), rows):
            want = np.nonzero(tp.lower[i, :j] & tp.lower[j, :j])[0].tolist()
            assert t["tile_k"][t["tile_kptr"][e]:t["tile_kptr"][e + 1]].tolist() == want


def test_rcm_ordering_makes_a_shuffled_chain_banded():
    from theseus_amd.sparse import TilePattern, rcm_order
    P = 600
    edges = chain_graph(P)
    blocks0 = np.array(sorted({(p, p) for p in range(P)} | {(max(a, b), min(a, b)) for a, b in edges}))
    perm = rcm_order(P, edges)
    pos = np.empty(P, dtype=np.int64)
    pos[perm] = np.arange(P)
    blocks1 = np.array(sorted({(p, p) for p in range(P)} | {(max(pos[a], pos[b]), min(pos[a], pos[b])) for a, b in edges}))
    t0, t1 = TilePattern(6 * P, blocks0, 6), TilePattern(6 * P, blocks1, 6)
    assert sorted(perm) == list(range(P))
    assert t1.tile_products * 8 < t0.tile_products             # shuffled labels: (almost) dense fill; RCM: a narrow band
    assert t1.tile_products * 10 < t1.dense_tile_products


@pytest.mark.parametrize("group", ["SE3", "SE2"])
def test_sparse_solver_equals_dense_solver_through_the_host_path(group):
    """LM on a shuffled chain graph: HipSparseCholeskySolver (fill-reducing VariableOrdering -> permuted packed poses, permuted
    Hessian columns, tile pattern) gives the dense solver's solution; the stand-in asserts that the numeric factor stays inside
    the symbolic pattern."""
    import theseus_amd as th
    from oracle import lie, lie_se2
    from tests.oracle_kernels import OracleKernels
    P, B, dtype = 90, 2, torch.float64
    edges = chain_graph(P, stride=5, span=4, seed=1)
    gen = torch.Generator().manual_seed(4)
    if group == "SE3":
        G, exp, comp, inv, dof = th.SE3, lie.se3_exp, lie.se3_compose, lie.se3_inverse, 6
    else:
        G, exp, comp, inv, dof = th.SE2, lie_se2.se2_exp, lie_se2.se2_compose, lie_se2.se2_inverse, 3
    gt = exp(1.5 * (2 * torch.rand(B * P, dof, dtype=dtype, generator=gen) - 1)).view(B, P, *G().tensor.shape[1:])
    noise = lambda n, s: exp(s * (2 * torch.rand(n, dof, dtype=dtype, generator=gen) - 1))  # noqa: E731
    rec = gt.shape[2:]
    poses0 = comp(gt.reshape(-1, *rec), noise(B * P, 0.05)).view(B, P, *rec)
    meas = [comp(comp(inv(gt[:, i]), gt[:, j]), noise(B, 0.01)) for (i, j) in edges]

    def run(solver_cls):
        obj = th.Objective(dtype=dtype)
        pv = [G(tensor=poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
        w = th.ScaleCostWeight(torch.tensor(5.0, dtype=dtype))
        for k, (i, j) in enumerate(edges):
            obj.add(th.Between(pv[i], pv[j], G(tensor=meas[k].clone(), name=f"m_{k}"), w, name=f"b_{k}"))
        obj.add(th.Difference(pv[edges[0][0]], G(tensor=gt[:, edges[0][0]].clone(), name="anchor"), w, name="prior"))
        opt = th.LevenbergMarquardt(obj, linear_solver_cls=solver_cls, linearization_kwargs=dict(kernels=OracleKernels()),
                                    max_iterations=5, abs_err_tolerance=0.0, rel_err_tolerance=0.0)
        sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(damping=1e-2, track_err_history=True))
        return torch.stack([sol[f"pose_{k}"] for k in range(P)], 1), info, opt
    dense, dinfo, _ = run(th.HipCholeskySolver)
    sparse, sinfo, opt = run(th.HipSparseCholeskySolver)
    lin = opt.linear_solver.linearization
    assert [v.name for v in lin.ordering] != [f"pose_{k}" for k in range(P)]          # a genuinely permuted column order
    assert opt.linear_solver.pattern.tile_products < opt.linear_solver.pattern.dense_tile_products
    np.testing.assert_allclose(sparse.numpy(), dense.numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(sinfo.err_history.numpy(), dinfo.err_history.numpy(), rtol=1e-9)
    assert sinfo.err_history[:, -1].mean() < 0.05 * sinfo.err_history[:, 0].mean()


def test_slot_tables_of_the_tile_packed_factor():
    """TilePattern's slot tables (include/theseus_hip.h: thx_tile_pattern, tile-packed factor): slot j = diagonal tile j, slot
    ntiles + e = off-diagonal entry e; every K-list element names the slots of its two operand tiles, every row-list element the
    slot of its tile -- and all of them are tiles of the pattern."""
    import numpy as np
    from theseus_amd.compiler import PoseGraphStructure
    from theseus_amd.sparse import tile_pattern
    s = PoseGraphStructure.build(400, chain_graph(400, stride=7, span=5, seed=2), [0])
    pat = tile_pattern(s, 6)
    t, nt = pat.tables, pat.ntiles
    assert pat.nslots == nt + len(t["col_row"]) == pat.l_tiles
    tiles = {v: k for k, v in pat.slot.items()}
    assert len(tiles) == pat.nslots and all(pat.lower[i, j] for (i, j) in pat.slot)
    for j in range(nt):
        for e in range(t["col_ptr"][j], t["col_ptr"][j + 1]):
            i = int(t["col_row"][e])
            assert pat.slot[(i, j)] == nt + e
            for q in range(t["tile_kptr"][e], t["tile_kptr"][e + 1]):
                k = int(t["tile_k"][q])
                assert tiles[int(t["tile_sa"][q])] == (j, k) and tiles[int(t["tile_sb"][q])] == (i, k)
        for q in range(t["diag_kptr"][j], t["diag_kptr"][j + 1]):
            assert tiles[int(t["diag_s"][q])] == (j, int(t["diag_k"][q]))
    for i in range(nt):
        for q in range(t["row_ptr"][i], t["row_ptr"][i + 1]):
            assert tiles[int(t["row_slot"][q])] == (i, int(t["row_tile"][q]))


# ---- elimination-tree parallelism: tile-level nested dissection + the level schedule (thx_chol_factor_levels) ----------------
def _level_setup(P=1200, seed=2, method="nd", batch_hint=64):
    from theseus_amd.compiler import PoseGraphStructure
    from theseus_amd.sparse import LevelPattern, tile_nested_dissection
    edges = chain_graph(P, stride=7, span=5, seed=seed)
    order, counts, info = tile_nested_dissection(P, edges, 128 // 6, batch_hint=batch_hint, method=method)
    pos = np.empty(P, dtype=np.int64)
    pos[np.asarray(order)] = np.arange(P)
    s = PoseGraphStructure.build(P, [(int(pos[a]), int(pos[b])) for a, b in edges], [0])
    return s, LevelPattern(s.lower_block_pattern(), 6, counts), info, order


def test_nested_dissection_of_a_chain_has_log_depth():
    from theseus_amd.sparse import nested_dissection, _symbolic_tiles
    nt = 200
    adj = np.zeros((nt, nt), dtype=bool)
    idx = np.arange(nt - 1)
    adj[idx, idx + 1] = adj[idx + 1, idx] = True
    order = np.asarray(nested_dissection(adj, leaf=1))
    assert sorted(order.tolist()) == list(range(nt))
    lp, level = _symbolic_tiles(np.tril(adj[np.ix_(order, order)]))
    assert level.max() + 1 <= 9                                  # ~log2(200) dependent levels instead of 200
    assert lp.sum() <= 3 * nt                                    # a node keeps at most its two separator neighbours: <= 3 tiles per column
    band, blevel = _symbolic_tiles(np.tril(adj))
    assert blevel.max() + 1 == nt and band.sum() == 2 * nt - 1   # the band order: no fill, a chain of nt dependent columns


def test_level_pattern_tables():
    s, pat, info, order = _level_setup()
    t, nt = pat.tables, pat.ntiles
    assert sorted(order) == list(range(1200)) and info["method"].startswith("nd")
    assert pat.tree_levels <= 8 and pat.tree_levels == info["levels"] and pat.nlevels <= 2 * pat.tree_levels and nt == (1200 + 20) // 21
    assert np.all(np.diff(pat.level) >= 0)                                         # block columns numbered level by level
    assert pat.level_col[0] == 0 and pat.level_col[-1] == nt and pat.level_ent[-1] == len(t["col_row"]) == pat.nslots - nt
    assert t["tile_valid"].tolist() == (6 * pat.tile_count).tolist() and t["tile_valid"].max() <= 128
    tiles = {v: k for k, v in pat.slot.items()}
    assert len(tiles) == pat.nslots == pat.l_tiles
    for lv in range(pat.nlevels):
        cols = range(pat.level_col[lv], pat.level_col[lv + 1])
        for j in cols:                                                              # a column depends on lower levels only
            dk = t["diag_k"][t["diag_kptr"][j]:t["diag_kptr"][j + 1]]
            assert dk.tolist() == np.nonzero(pat.lower[j, :j])[0].tolist() and all(pat.level[k] < lv for k in dk)
            assert [tiles[int(q)] for q in t["diag_s"][t["diag_kptr"][j]:t["diag_kptr"][j + 1]]] == [(j, int(k)) for k in dk]
        ents = range(pat.level_ent[lv], pat.level_ent[lv + 1])
        assert sorted((int(t["col_row"][e]), int(t["ent_col"][e])) for e in ents) == \
            sorted((int(i), j) for j in cols for i in np.nonzero(pat.lower[j + 1:, j])[0] + j + 1)
        klens = [int(t["tile_kptr"][e + 1] - t["tile_kptr"][e]) for e in ents]
        assert klens == sorted(klens, reverse=True)                                 # longest K-list first inside a level
        for e in ents:
            i, j = int(t["col_row"][e]), int(t["ent_col"][e])
            q0, q1 = t["tile_kptr"][e], t["tile_kptr"][e + 1]
            assert t["tile_k"][q0:q1].tolist() == np.nonzero(pat.lower[i, :j] & pat.lower[j, :j])[0].tolist()
            assert pat.slot[(i, j)] == nt + e
            for q in range(q0, q1):
                k = int(t["tile_k"][q])
                assert tiles[int(t["tile_sa"][q])] == (j, k) and tiles[int(t["tile_sb"][q])] == (i, k)
    # rows of one level touch disjoint column blocks (the backward solve pushes without atomics)
    for lv in range(pat.nlevels):
        seen = set()
        for i in range(pat.level_col[lv], pat.level_col[lv + 1]):
            ks = set(t["row_tile"][t["row_ptr"][i]:t["row_ptr"][i + 1]].tolist())
            assert not (ks & seen)
            seen |= ks
    # padded order <-> column order
    assert pat.npad == 128 * nt and pat.n == 7200
    assert np.array_equal(pat.col_of_pad[pat.pad_of_col], np.arange(pat.n)) and (pat.col_of_pad < 0).sum() == pat.npad - pat.n


def test_the_time_model_keeps_the_band_order_for_huge_batches():
    """At thousands of problems per call the batch alone fills the chip: an order with the least arithmetic wins (the band, or
    the two half-chains running towards one separator -- the "twisted" factorisation: the band's tile count at half its depth);
    at the reference's sweep sizes (evaluations/pose_graph_synthetic.sh: batch 8 - 256) the log-depth order does."""
    from theseus_amd.sparse import tile_nested_dissection
    edges = chain_graph(2048, stride=7, span=5, seed=2)
    big = tile_nested_dissection(2048, edges, 21, batch_hint=8192)[2]
    assert big["l_tiles"] == big["candidates"]["band"]["l_tiles"] and big["levels"] <= big["candidates"]["band"]["levels"]
    small = tile_nested_dissection(2048, edges, 21, batch_hint=64)[2]
    assert small["method"].startswith("nd") and small["levels"] <= 12


def test_level_schedule_emulated_on_the_host_solves_the_system():
    """The level schedule executed literally -- tile by tile from the device tables, every level's columns / entries / rows in
    ARBITRARY order (they must not depend on each other), H gathered through the padded piece tables, vectors through the
    gather maps -- gives the solution of the unpadded system."""
    s, pat, _, _ = _level_setup(P=500, seed=5)
    hb = s.hessian_blocks()
    rng = np.random.default_rng(0)
    n, d, T, nt = pat.n, 6, 128, pat.ntiles
    A = np.zeros((n, n))
    for r, c in s.lower_block_pattern():
        A[d * r:d * r + d, d * c:d * c + d] = rng.standard_normal((d, d))
    A = np.tril(A) + np.tril(A, -1).T + 30 * np.eye(n)
    Hc = hb.pack_dense(A[None])[0]
    tile_ptr, piece_blk, piece_rc = pat.piece_tables(hb)

    def h_tile(i, j):
        out = np.zeros((T, T))
        tt = i * (i + 1) // 2 + j
        for pc in range(tile_ptr[tt], tile_ptr[tt + 1]):
            rc = int(piece_rc[pc]) & 0xFFFFFFFF
            r0, c0 = rc >> 16, rc & 0xFFFF
            out[r0:r0 + d, c0:c0 + d] = Hc[piece_blk[pc] * d * d:(piece_blk[pc] + 1) * d * d].reshape(d, d)
        return out
    t = pat.tables
    L = np.zeros((pat.nslots, T, T))
    for lv in range(pat.nlevels):
        for j in rng.permutation(np.arange(pat.level_col[lv], pat.level_col[lv + 1])):
            S = np.tril(h_tile(j, j))
            S = S + np.tril(S, -1).T
            v = int(t["tile_valid"][j])
            S[v:, :] = 0
            S[:, v:] = 0
            S[np.arange(v, T), np.arange(v, T)] = 1.0                               # identity padding
            for q in range(t["diag_kptr"][j], t["diag_kptr"][j + 1]):
                S -= L[t["diag_s"][q]] @ L[t["diag_s"][q]].T
            L[j] = np.linalg.cholesky(S)
        for e in rng.permutation(np.arange(pat.level_ent[lv], pat.level_ent[lv + 1])):
            i, j = int(t["col_row"][e]), int(t["ent_col"][e])
            Pm = h_tile(i, j)
            for q in range(t["tile_kptr"][e], t["tile_kptr"][e + 1]):
                Pm -= L[t["tile_sb"][q]] @ L[t["tile_sa"][q]].T
            L[nt + e] = np.linalg.solve(L[j], Pm.T).T
    g = rng.standard_normal(n)
    y = np.zeros(pat.npad)
    y[pat.pad_of_col] = g
    for lv in range(pat.nlevels):                                                   # forward, bottom up: rows pull
        for i in rng.permutation(np.arange(pat.level_col[lv], pat.level_col[lv + 1])):
            acc = y[T * i:T * i + T].copy()
            for q in range(t["row_ptr"][i], t["row_ptr"][i + 1]):
                k = int(t["row_tile"][q])
                acc -= L[t["row_slot"][q]] @ y[T * k:T * k + T]
            y[T * i:T * i + T] = np.linalg.solve(L[i], acc)
    for lv in reversed(range(pat.nlevels)):                                         # backward, top down: rows push
        for i in rng.permutation(np.arange(pat.level_col[lv], pat.level_col[lv + 1])):
            xi = np.linalg.solve(L[i].T, y[T * i:T * i + T])
            y[T * i:T * i + T] = xi
            for q in range(t["row_ptr"][i], t["row_ptr"][i + 1]):
                k = int(t["row_tile"][q])
                y[T * k:T * k + T] -= L[t["row_slot"][q]].T @ xi
    x = y[pat.pad_of_col]
    np.testing.assert_allclose(x, np.linalg.solve(A, g), rtol=0, atol=1e-11)
    assert np.abs(y[pat.col_of_pad < 0]).max() == 0.0                               # padding stays exactly zero


def test_subtree_groups_of_the_tile_elimination_tree():
    """Stream groups for thx_level_schedule.level_stream_host: a band is one chain (no split); two chains towards a separator are
    groups 0 / 1 with the separator's chain as the trunk (-1); a subtree never shares a K-list with the other group."""
    from theseus_amd.sparse import _subtree_groups, _symbolic_tiles
    nt = 9
    band = np.eye(nt, dtype=bool) | np.eye(nt, k=-1, dtype=bool)
    assert _subtree_groups(_symbolic_tiles(band)[0]).tolist() == [0] * nt
    # columns 0-2 and 3-5: two chains; 6-8: the separator chain both end in
    lp0 = np.eye(nt, dtype=bool)
    for a, b in ((1, 0), (2, 1), (4, 3), (5, 4), (6, 2), (6, 5), (7, 6), (8, 7)):
        lp0[a, b] = True
    lp, level = _symbolic_tiles(lp0)
    g = _subtree_groups(lp)
    assert g[6:].tolist() == [-1, -1, -1] and sorted({tuple(g[:3].tolist()), tuple(g[3:6].tolist())}) == [(0, 0, 0), (1, 1, 1)]
    for j in range(6):                                   # a column's K-list (its row of L) stays inside its own group
        assert all(g[k] == g[j] for k in np.nonzero(lp[j, :j])[0])
