"""ctypes binding of libtheseus_hip.so (C ABI: include/theseus_hip.h).

There is NO fallback: if the shared library is missing or a tensor is not on a HIP device the
calls raise.  ``import torch`` happens first on purpose, so that the HIP runtime torch already
loaded (same SONAME libamdhip64.so.7) is the one our kernels are registered with -- device
pointers and streams are then shared with torch.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# THESEUS_HIP_LIB overrides the in-tree build (kernel A/B experiments: tools/variants.sh)
LIB_PATH = os.environ.get("THESEUS_HIP_LIB") or os.path.join(_HERE, "lib", "libtheseus_hip.so")

THX_TILE = 128
THX_ERR_CHUNKS = 128
THX_BA_ERR_CHUNKS = 256
LOSS_NONE, LOSS_WELSCH, LOSS_HUBER, LOSS_HINGE = 0, 1, 2, 3  # THX_LOSS_* (theseus/core/robust_loss.py:33-62)
LOSS_FLATTEN = 4  # THX_LOSS_FLATTEN: RobustCostFunction(flatten_dims=True), or-ed into a loss code
LOSS_GEMAN_MCCLURE = 8  # THX_LOSS_GEMAN_MCCLURE (robust_loss.py:92-113; the radius entry carries log(mu * radius))
ABI_VERSION = 24


class LieEps(Structure):
    _fields_ = [("near_zero", c_double), ("d_near_zero", c_double), ("near_pi", c_double)]


class BAStructure(Structure):  # thx_ba_structure: int32 device tables of a bundle-adjustment objective
    _fields_ = [(k, c_int32) for k in ("num_cams", "num_points", "num_obs", "num_cam_priors", "num_pt_priors", "num_pairs", "num_blocks")] + [
        (k, c_void_p) for k in ("obs_cam", "obs_pt", "pt_ptr", "pt_obs", "cam_ptr", "cam_obs", "cam_prior_cam",
                                "cam_prior_ptr", "cam_prior_id", "pt_prior_pt", "pt_prior_ptr", "pt_prior_id",
                                "pair_ptr", "pair_o1", "pair_o2", "pair_c2", "pair_dptr", "blk_ptr", "blk_c1", "blk_c2")]


class BAData(Structure):  # thx_ba_data
    _fields_ = [
        ("batch", c_int32), ("cams", c_void_p), ("points", c_void_p),
        ("feat", c_void_p), ("feat_bstride", c_int64), ("w_obs", c_void_p), ("w_obs_bstride", c_int64),
        ("focal", c_void_p), ("k1", c_void_p), ("k2", c_void_p), ("calib_bstride", c_int64),
        ("robust_obs", c_int32), ("log_radius_obs", c_void_p), ("log_radius_obs_bstride", c_int64),
        ("cam_prior_target", c_void_p), ("cam_prior_target_bstride", c_int64),
        ("w_cam_prior", c_void_p), ("w_cam_prior_bstride", c_int64),
        ("pt_prior_target", c_void_p), ("pt_prior_target_bstride", c_int64),
        ("w_pt_prior", c_void_p), ("w_pt_prior_bstride", c_int64),
    ]


BA_UNROLL_GRADS = ("cam_obs", "pt_obs", "feat", "w_obs", "focal", "k1", "k2", "log_radius_obs", "cam_prior_cam", "cam_prior_target",
                   "w_cam_prior", "pt_prior_pt", "pt_prior_target", "w_pt_prior")


class BAUnrollGrads(Structure):  # thx_ba_unroll_grads: per-cost outputs of thx_ba_unroll_vjp
    _fields_ = [(k, c_void_p) for k in BA_UNROLL_GRADS]


class TilePattern(Structure):  # thx_tile_pattern: tile-level symbolic factorisation (device int32 tables + one host table)
    _fields_ = [("ntiles", c_int32)] + [(k, c_void_p) for k in ("col_ptr", "col_row", "tile_kptr", "tile_k", "diag_kptr", "diag_k",
                                                               "col_count_host", "row_ptr", "row_tile",
                                                               "tile_sa", "tile_sb", "diag_s", "row_slot")] + [("nslots", c_int32), ("col_head_host", c_void_p)]


class CholSchedule(Structure):  # thx_chol_schedule: per-call schedule of the factorisations (negative = the library default)
    _fields_ = [("split_diag_min_batch", c_int32), ("column_pairs", c_int32), ("right_looking_max_batch", c_int32),
                ("hb_scatter_max_pieces", c_int32), ("f64_wide_max_ktiles", c_int32),
                ("f64_half_max_ktiles", c_int32)]


class LevelSchedule(Structure):  # thx_level_schedule: elimination-tree levels of a tile pattern (2 host + 2 device int32 tables)
    _fields_ = [("nlevels", c_int32)] + [(k, c_void_p) for k in ("level_col_host", "level_ent_host", "level_maxk_host", "ent_col", "tile_valid",
                                                                 "level_stream_host")]


class HBlockLayout(Structure):  # thx_hblock_layout: block-compact Hessian (device int32 tables)
    _fields_ = [(k, c_int32) for k in ("nblocks", "bd", "nvars", "ntiles")] + [
        (k, c_void_p) for k in ("diag_blk", "inc_blk", "tile_ptr", "piece_blk", "piece_rc")] + [("max_tile_pieces", c_int32)]


def max_offdiag_tile_pieces(tile_ptr, ntiles: int) -> int:
    """thx_hblock_layout.max_tile_pieces: the largest piece count of a lower tile t = i (i + 1) / 2 + j with i > j (0: none)."""
    import numpy as np
    cnt = np.diff(np.asarray(tile_ptr, dtype=np.int64))
    if cnt.size < ntiles * (ntiles + 1) // 2 or ntiles < 2:
        return 0
    off = np.ones(ntiles * (ntiles + 1) // 2, bool)
    i = np.arange(ntiles)
    off[i * (i + 1) // 2 + i] = False
    return int(cnt[: off.size][off].max())


class SE2Eps(Structure):  # thx_se2_eps (theseus/global_params.py:46-59)
    _fields_ = [("near_zero", c_double), ("d_near_zero", c_double)]


class PGStructure(Structure):
    _fields_ = [
        ("num_poses", c_int32), ("num_edges", c_int32), ("num_priors", c_int32),
        ("edge_i", c_void_p), ("edge_j", c_void_p),
        ("inc_ptr", c_void_p), ("inc_edge", c_void_p), ("inc_side", c_void_p), ("inc_other", c_void_p),
        ("prior_pose", c_void_p), ("pri_ptr", c_void_p), ("pri_id", c_void_p),
    ]


class PGData(Structure):
    _fields_ = [
        ("batch", c_int32),
        ("poses", c_void_p),
        ("meas", c_void_p), ("meas_bstride", c_int64),
        ("w_between", c_void_p), ("w_between_bstride", c_int64),
        ("prior_target", c_void_p), ("prior_target_bstride", c_int64),
        ("w_prior", c_void_p), ("w_prior_bstride", c_int64),
        # RobustCostFunction wrappers: loss code (LOSS_* [| LOSS_FLATTEN]) per cost role + log_loss_radius (E|K, Br, 1);
        # loss_<role>: int32 code per cost when the costs of a role differ
        ("robust_between", c_int32), ("log_radius_between", c_void_p), ("log_radius_between_bstride", c_int64),
        ("robust_prior", c_int32), ("log_radius_prior", c_void_p), ("log_radius_prior_bstride", c_int64),
        ("loss_between", c_void_p), ("loss_prior", c_void_p),
    ]


# name -> (argtypes) ; every entry point returns int (0 = ok)
_SIGNATURES = {
    "thx_se3_exp": [c_void_p, c_void_p, c_void_p, c_int64, c_int, POINTER(LieEps), c_void_p],
    "thx_se3_log": [c_void_p, c_void_p, c_void_p, c_int64, c_int, POINTER(LieEps), c_void_p],
    "thx_se3_compose": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p],
    "thx_se3_inverse": [c_void_p, c_void_p, c_int64, c_int, c_void_p],
    "thx_se3_adjoint": [c_void_p, c_void_p, c_int64, c_int, c_void_p],
    "thx_pg_assemble": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_int64, c_void_p, c_int,
                        POINTER(LieEps), c_void_p],
    "thx_pg_error": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_void_p, c_int, POINTER(LieEps), c_void_p],
    "thx_pg_jacobians": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_void_p, c_void_p, c_void_p,
                         c_void_p, c_int, POINTER(LieEps), c_void_p],
    "thx_se3_retract": [c_void_p, c_void_p, c_int64, c_double, c_void_p, c_void_p, c_int32, c_int32, c_int,
                        POINTER(LieEps), c_void_p],
    "thx_pg2_assemble": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_int64, c_void_p, c_int, POINTER(SE2Eps),
                         c_void_p],
    "thx_pg2_error": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_void_p, c_int, POINTER(SE2Eps), c_void_p],
    "thx_pg2_jacobians": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                          POINTER(SE2Eps), c_void_p],
    "thx_se2_retract": [c_void_p, c_void_p, c_int64, c_double, c_void_p, c_void_p, c_int32, c_int32, c_int,
                        POINTER(SE2Eps), c_void_p],
    "thx_se2_op": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, POINTER(SE2Eps), c_void_p],
    "thx_pgso3_assemble": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_int64, c_void_p, c_int, POINTER(LieEps),
                           c_void_p],
    "thx_pgso3_error": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_void_p, c_int, POINTER(LieEps), c_void_p],
    "thx_pgso3_jacobians": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                            POINTER(LieEps), c_void_p],
    "thx_so3_retract": [c_void_p, c_void_p, c_int64, c_double, c_void_p, c_void_p, c_int32, c_int32, c_int,
                        POINTER(LieEps), c_void_p],
    "thx_so3_op": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, POINTER(LieEps), c_void_p],
    "thx_pgso2_assemble": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_int64, c_void_p, c_int, c_void_p],
    "thx_pgso2_error": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_void_p, c_int, c_void_p],
    "thx_pgso2_jacobians": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "thx_so2_retract": [c_void_p, c_void_p, c_int64, c_double, c_void_p, c_void_p, c_int32, c_int32, c_int, c_void_p],
    "thx_so2_op": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p],
    "thx_ba_assemble": [POINTER(BAStructure), POINTER(BAData), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                        c_int, POINTER(LieEps), c_void_p],
    "thx_ba_schur": [POINTER(BAStructure), c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_double,
                     c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "thx_ba_schur_blocks": [POINTER(BAStructure), c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_double,
                            c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "thx_ba_backsub": [POINTER(BAStructure), c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p],
    "thx_ba_error": [POINTER(BAStructure), POINTER(BAData), c_void_p, c_void_p, c_int, POINTER(LieEps), c_void_p],
    "thx_ba_av": [POINTER(BAStructure), POINTER(BAData), c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                  POINTER(LieEps), c_void_p],
    "thx_ba_vjp": [POINTER(BAStructure), POINTER(BAData), c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, POINTER(LieEps), c_void_p],
    "thx_ba_unroll_vjp": [POINTER(BAStructure), POINTER(BAData), c_void_p, c_int64, c_void_p, c_int64, c_void_p, POINTER(BAUnrollGrads),
                          c_int, POINTER(LieEps), c_void_p],
    "thx_copy_where": [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p],
    "thx_vec_retract": [c_void_p, c_void_p, c_int64, c_int64, c_double, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int,
                        c_void_p],
    "thx_lm_accept_diag": [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int,
                           c_double, c_double, c_double, c_void_p, c_int, c_void_p],
    "thx_se2_retract_vjp": [c_void_p, c_void_p, c_int64, c_double, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int,
                            POINTER(SE2Eps), c_void_p],
    "thx_pg2_vjp": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                    c_void_p, c_void_p, c_int, POINTER(SE2Eps), c_void_p],
    "thx_so3_retract_vjp": [c_void_p, c_void_p, c_int64, c_double, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int,
                            POINTER(LieEps), c_void_p],
    "thx_pgso3_vjp": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                      c_void_p, c_void_p, c_int, POINTER(LieEps), c_void_p],
    "thx_chol_factor": [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int, c_double, c_void_p, c_void_p,
                        c_void_p, c_int, c_void_p, POINTER(CholSchedule)],
    "thx_chol_factor_forward": [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int, c_double, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, POINTER(CholSchedule)],
    "thx_chol_factor_sparse": [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_int64, POINTER(TilePattern), c_int, c_void_p, POINTER(CholSchedule)],
    "thx_chol_solve_sparse": [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                              POINTER(TilePattern), c_int, c_void_p],
    "thx_pg_assemble_blocks": [POINTER(PGStructure), POINTER(PGData), POINTER(HBlockLayout), c_void_p, c_int64, c_void_p, c_int,
                               POINTER(LieEps), c_void_p],
    "thx_hblocks_expand": [POINTER(HBlockLayout), c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int, c_void_p],
    "thx_hblocks_diag": [POINTER(HBlockLayout), c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int, c_void_p],
    "thx_chol_factor_hblocks": [POINTER(HBlockLayout), c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int, c_double, c_void_p,
                                c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, POINTER(TilePattern), c_int, c_void_p,
                                POINTER(CholSchedule)],
    "thx_chol_factor_levels": [POINTER(HBlockLayout), c_void_p, c_int64, c_int32, c_void_p, c_int, c_double, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_int64, POINTER(TilePattern), POINTER(LevelSchedule), c_int, c_void_p,
                               POINTER(CholSchedule)],
    "thx_chol_solve_levels": [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int, POINTER(TilePattern),
                              POINTER(LevelSchedule), c_int, c_void_p],
    "thx_vec_gather": [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int, c_void_p],
    "thx_chol_solve": [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                       c_void_p],
    "thx_chol_solve_backward": [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                c_void_p],
    "thx_se3_retract_vjp": [c_void_p, c_void_p, c_int64, c_double, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int,
                            POINTER(LieEps), c_void_p],
    "thx_pg_vjp": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                   c_void_p, c_void_p, c_int, POINTER(LieEps), c_void_p],
    "thx_pg_unroll_vjp": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, POINTER(LieEps), c_void_p],
    "thx_pg2_unroll_vjp": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, POINTER(SE2Eps), c_void_p],
    "thx_pgso3_unroll_vjp": [POINTER(PGStructure), POINTER(PGData), c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, POINTER(LieEps), c_void_p],
    "thx_block_assemble": [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int64,
                           c_void_p, c_int64, c_int32, c_int, c_void_p],
    "thx_diag": [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int64, c_int, c_void_p],
    "thx_lm_accept": [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                      c_void_p, c_int, c_double, c_double, c_double, c_void_p, c_int, c_void_p],
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["thx_last_error", "thx_abi_version"])

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built.  Run "
            "`python -m theseus_amd.build` (hipcc, gfx950).  There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.thx_last_error.restype = c_char_p
    lib.thx_last_error.argtypes = []
    lib.thx_abi_version.restype = c_int
    lib.thx_abi_version.argtypes = []
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = c_int
        fn.argtypes = args
    if lib.thx_abi_version() != ABI_VERSION:
        raise RuntimeError("libtheseus_hip.so ABI version mismatch: rebuild with python -m theseus_amd.build")
    _lib = lib
    return lib


_TRACE_CALLS = bool(int(os.environ.get("THX_TRACE_CALLS", "0")))   # debugging aid: synchronise + name every library call


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {load().thx_last_error().decode()}")
    if _TRACE_CALLS:
        import sys
        print(f"[thx] {what} queued", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        print(f"[thx] {what} done", file=sys.stderr, flush=True)


def dtype_code(dtype):
    if dtype == torch.float32:
        return 0
    if dtype == torch.float64:
        return 1
    raise TypeError(f"theseus_amd HIP kernels support float32/float64, got {dtype}")


def ptr(t, name="tensor"):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device (got {t.device}); there is no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
