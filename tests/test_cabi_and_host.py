"""CPU-side (-m "not gpu") checks: the C-ABI library builds, loads and exports every symbol
include/theseus_hip.h declares; host-side structure compiler and batch sharding helpers."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT


@pytest.fixture(scope="session")
def lib_path():
    from theseus_amd import build
    return build.build(verbose=False)  # hipcc cross-compiles gfx950 without a GPU; no-op when up to date


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "theseus_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(thx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib_path):
    names = declared_symbols()
    assert len(names) >= 15 and "thx_chol_factor_forward" in names and "thx_pg_assemble" in names
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/theseus_hip.h but not exported: {missing}"


def test_ctypes_binding_covers_the_header(lib_path):
    from theseus_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()
    lib = _lib.load()
    assert lib.thx_abi_version() == _lib.ABI_VERSION
    assert lib.thx_last_error() is not None


def test_bad_arguments_are_rejected_without_a_gpu(lib_path):
    """Argument validation happens on the host side of the C ABI, before any launch."""
    from theseus_amd import _lib
    lib = _lib.load()
    rc = lib.thx_chol_factor(None, 32, 6, 1, None, 0, 1e-8, None, None, None, 0, None, None)
    assert rc != 0 and b"null pointer" in lib.thx_last_error()
    one = ctypes.c_void_p(16)
    rc = lib.thx_chol_factor(one, 33, 6, 1, None, 0, 1e-8, one, one, one, 0, None, None)  # ld % 32 != 0
    assert rc != 0 and b"ld" in lib.thx_last_error()
    rc = lib.thx_chol_factor(one, 32, 6, 1, None, 0, 1e-8, one, one, one, 7, None, None)  # bad dtype
    assert rc != 0 and b"dtype" in lib.thx_last_error()
    # round 6 entry points: thx_ba_schur_blocks (block list too short / no block tables), thx_vec_gather (aliasing), the level pair
    s = _lib.BAStructure()
    s.num_cams, s.num_points, s.num_blocks = 4, 8, 3
    args = lambda sc, bstride, diag, dst: (ctypes.byref(s), 2, one, one, one, one, 48, None, 0, 1e-8, sc, bstride, diag, dst, one, 24,  # noqa: E731
                                           one, one, one, 0, None)
    rc = lib.thx_ba_schur_blocks(*args(one, 36 * 7, None, one))
    assert rc != 0 and b"null argument" in lib.thx_last_error()
    rc = lib.thx_ba_schur_blocks(*args(one, 36 * 7, one, None))
    assert rc != 0 and b"blk_dst" in lib.thx_last_error()
    rc = lib.thx_ba_schur_blocks(*args(one, 36 * 6, one, one))        # 4 diagonal + 3 pair blocks need 36 * 7 elements
    assert rc != 0 and b"bstride" in lib.thx_last_error()
    rc = lib.thx_vec_gather(one, 8, one, 8, one, 8, 2, 0, None)          # src == dst
    assert rc != 0 and b"bad args" in lib.thx_last_error()
    rc = lib.thx_chol_factor_levels(None, one, 36, 2, None, 0, 1e-8, one, one, one, None, None, 0, None, None, 0, None, None)
    assert rc != 0 and b"block layout" in lib.thx_last_error()
    sched = _lib.CholSchedule(-1, -1, -1, -1, -1, -1)                               # (a schedule does not rescue bad arguments)
    rc = lib.thx_chol_factor(one, 33, 6, 1, None, 0, 1e-8, one, one, one, 0, None, ctypes.byref(sched))
    assert rc != 0 and b"ld" in lib.thx_last_error()


def test_product_has_no_cpu_fallback():
    """Tensors that are not on a HIP device are refused loudly (theseus_amd/_lib.py:ptr)."""
    import torch
    from theseus_amd import _lib
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.ptr(torch.zeros(4), "x")


def test_structure_compiler_matches_reference_layout():
    from theseus_amd.compiler import PoseGraphStructure
    edges = [(0, 1), (1, 2), (2, 0), (3, 1), (0, 3)]
    s = PoseGraphStructure.build(4, edges, [0, 2])
    assert s.num_cols == 24 and s.num_rows == 6 * 7 and s.var_start_cols == [0, 6, 12, 18]
    # incident-edge CSR: pose 0 touches edges 0 (v0), 2 (v1), 4 (v0); sorted by the other endpoint
    lo, hi = s.inc_ptr[0], s.inc_ptr[1]
    assert s.inc_other[lo:hi].tolist() == [1, 2, 3]
    assert s.inc_edge[lo:hi].tolist() == [0, 2, 4] and s.inc_side[lo:hi].tolist() == [0, 1, 0]
    assert s.lower_block_pattern().tolist() == [[0, 0], [1, 0], [1, 1], [2, 0], [2, 1], [2, 2], [3, 0], [3, 1], [3, 3]]
    with pytest.raises(ValueError):
        PoseGraphStructure.build(3, [(0, 0)], [])
    with pytest.raises(ValueError):
        PoseGraphStructure.build(3, [(0, 5)], [])


def test_shard_bounds_cover_the_batch():
    from theseus_amd.sharding import shard_bounds
    for total in (1, 7, 8, 4096, 32768 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def test_plan_sub_batches_strong_scaling():
    """bench.py --total-batch: configs[2] = 32768 problems over 1/2/4/8 ranks in sub-batches of 4096."""
    import pytest
    from theseus_amd.sharding import plan_sub_batches, shard_bounds
    for world in (1, 2, 4, 8):
        plans = [plan_sub_batches(32768, r, world, 4096) for r in range(world)]
        assert plans == [(4096, 8 // world)] * world
        assert sum(b * n for b, n in plans) == 32768
    assert plan_sub_batches(8192, 0, 1, 4096) == (4096, 2)
    assert plan_sub_batches(1000, 1, 2, 4096) == (500, 1)          # a share below the sub-batch is one batch
    assert plan_sub_batches(10, 2, 3, 4) == (3, 1) and shard_bounds(10, 2, 3) == (7, 10)
    with pytest.raises(ValueError):
        plan_sub_batches(12288 + 8, 0, 1, 4096)                    # share not a multiple of the sub-batch
    with pytest.raises(ValueError):
        plan_sub_batches(2, 2, 3, 4096)                            # more ranks than problems


def test_integration_doc_lists_every_entry_point():
    """INTEGRATION.md is the binding a maintainer reads: every function include/theseus_hip.h declares appears in it."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "theseus_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(thx_[a-z0-9_]+)\s*\(", header)))
    assert len(names) > 30
    missing = [n for n in names if n not in doc]
    assert not missing, missing


def test_hessian_block_layout_round_trip():
    """theseus_amd/compiler.py:HessianBlocks -- the block-compact layout of tril(H) the device kernels read: every lower tile's
    piece list, applied the way the kernels gather it, reproduces the dense lower frame (blocks straddling tile boundaries
    included), for dof 6 and 3, natural and permuted variable orders."""
    import numpy as np
    from theseus_amd.compiler import PoseGraphStructure
    from theseus_amd.utils.synthetic import pose_graph_topology
    rng = np.random.default_rng(0)
    for P, E, d, shuffle in ((256, 1024, 6, False), (70, 200, 6, True), (50, 120, 3, False)):
        edges = pose_graph_topology(P, E, 0)
        if shuffle:
            perm = rng.permutation(P)
            edges = [(int(perm[i]), int(perm[j])) for i, j in edges]
        s = PoseGraphStructure.build(P, edges, [0], dof=d)
        hb = s.hessian_blocks()
        assert hb.nblocks == P + len({(max(i, j), min(i, j)) for i, j in edges})
        assert (hb.inc_blk >= 0).sum() == len(edges) and hb.bstride % 4 == 0 and hb.bstride >= hb.nblocks * d * d
        n = d * P
        ld = (n + 31) // 32 * 32
        H = np.zeros((2, ld, ld))
        for a, b in hb.blocks.tolist():
            H[:, d * a:d * a + d, d * b:d * b + d] = rng.normal(size=(2, d, d))
        tiles = np.zeros((ld, ld), dtype=bool)
        for ti in range(hb.ntiles):
            for tj in range(ti + 1):
                tiles[128 * ti:128 * ti + 128, 128 * tj:128 * tj + 128] = True
        assert np.array_equal(hb.expand(hb.pack_dense(H), ld) * tiles, H * tiles)
        # the blocks of a tile form one contiguous run (top-left ownership): ids ascend tile by tile
        own = [(d * a // 128, d * b // 128) for a, b in hb.blocks.tolist()]
        assert own == sorted(own)


def test_max_tile_pieces_and_schedule_structs_follow_the_header():
    """thx_hblock_layout.max_tile_pieces (host-computed: picks how the off-diagonal Cholesky kernels take their pieces of H) and the
    ctypes mirrors of the two structs that grew this round: the field lists are the header's, in order."""
    import re
    import numpy as np
    from theseus_amd import _lib
    from theseus_amd.compiler import PoseGraphStructure
    from theseus_amd.utils.synthetic import pose_graph_topology
    # the largest piece count over the OFF-diagonal lower tiles t = i (i + 1) / 2 + j, i > j
    tile_ptr = np.cumsum([0, 50, 3, 60, 7, 9, 70])          # tiles (0,0) (1,0) (1,1) (2,0) (2,1) (2,2)
    assert _lib.max_offdiag_tile_pieces(tile_ptr, 3) == 9
    assert _lib.max_offdiag_tile_pieces(np.array([0, 5]), 1) == 0     # one tile: no off-diagonal tile
    # the headline graph: a few tens of blocks per tile at most -> the matrix-core scatter (<= 64); equals a direct count
    s = PoseGraphStructure.build(256, pose_graph_topology(256, 1024, 0), [0], dof=6)
    hb = s.hessian_blocks()
    cnt = np.diff(hb.tile_ptr)
    off = [int(cnt[i * (i + 1) // 2 + j]) for i in range(hb.ntiles) for j in range(i)]
    assert _lib.max_offdiag_tile_pieces(hb.tile_ptr, hb.ntiles) == max(off) and 1 <= max(off) <= 64
    header = open(os.path.join(ROOT, "include", "theseus_hip.h")).read()

    def fields(name):
        body = re.search(r"typedef struct \{([^}]*)\} " + name + ";", header).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return [re.findall(r"\w+", piece)[-1] for decl in body.split(";") if decl.strip() for piece in decl.split(",")]
    assert fields("thx_chol_schedule") == [f for f, _ in _lib.CholSchedule._fields_]
    assert fields("thx_hblock_layout") == [f for f, _ in _lib.HBlockLayout._fields_]
