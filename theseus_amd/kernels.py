"""Kernel backend: torch tensors in, HIP launches out (through the C ABI, on torch's current stream).

``HipKernels`` is the only backend the product ships.  The class boundary exists so that host logic
that does not depend on a GPU (batch sharding, LM control flow) can be unit-tested on CPU with a
stand-in injected by tests/; nothing in this package provides such a stand-in.
"""
import dataclasses
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from .compiler import DeviceStructure

# ---- Taylor-switch thresholds and the one behavioural switch of the reference's global parameters ------------------------
# torchlie/torchlie/global_params.py:44-58 and theseus/global_params.py:46-80 (defaults).  They are read AT EVERY LAUNCH:
#   * stand-alone (this package's own mirror classes): from the tables below, set through ``set_global_params`` with the
#     reference's option names (``so3_near_zero_eps_float32`` ..., ``se2_near_zero_eps_float64`` ..., ``fast_approx_local_jacobians``);
#   * plugged into the real ``theseus`` (theseus_amd/plugin.py): from the reference's own ``_THESEUS_GLOBAL_PARAMS`` /
#     ``_TORCHLIE_GLOBAL_PARAMS`` objects (``use_reference_global_params``), so that ``torchlie.set_global_params`` /
#     ``theseus.set_global_params`` calls made by the user apply to the HIP kernels too.
_LIE_EPS = {
    torch.float32: dict(near_zero=1e-2, d_near_zero=2e-1, near_pi=1e-2),
    torch.float64: dict(near_zero=5e-3, d_near_zero=1e-2, near_pi=1e-7),
}
_SE2_EPS = {
    torch.float32: dict(near_zero=3e-2, d_near_zero=1e-1),
    torch.float64: dict(near_zero=1e-6, d_near_zero=1e-3),
}
_FLAGS = dict(fast_approx_local_jacobians=False)
_REFERENCE_PARAMS = None   # the reference's _THESEUS_GLOBAL_PARAMS (its get_eps falls through to torchlie's for so3_*)


def use_reference_global_params(theseus_params) -> None:
    global _REFERENCE_PARAMS
    _REFERENCE_PARAMS = theseus_params


def _option(key: str):
    """'so3_d_near_zero_eps_float32' -> (table, dtype, 'd_near_zero')."""
    for group, table in (("so3", _LIE_EPS), ("se2", _SE2_EPS)):
        for dt, tag in ((torch.float32, "float32"), (torch.float64, "float64")):
            if key.startswith(group + "_") and key.endswith("_eps_" + tag):
                attr = key[len(group) + 1:-len("_eps_" + tag)]
                if attr in table[dt]:
                    return table, dt, attr
    return None


def _forward_to_reference(options) -> None:
    """theseus_amd.plugin is loaded: every launch reads the REFERENCE's parameter objects, so a setter of this package must
    reach them (the local tables are no longer read -- silently updating them would make the call a no-op)."""
    import theseus
    import torchlie
    lie_opts = {k: v for k, v in options.items() if k.startswith("so3_")}
    rest = {k: v for k, v in options.items() if not k.startswith("so3_")}
    if lie_opts:
        torchlie.set_global_params(lie_opts)
    if rest:
        theseus.set_global_params(rest)


def set_global_params(options) -> None:
    """Mirror of ``theseus.set_global_params`` + ``torchlie.set_global_params`` for the options this path reads.
    Once ``theseus_amd.plugin`` redirected the look-ups to the reference's own objects the call is forwarded there."""
    if _REFERENCE_PARAMS is not None:
        for k in options:
            if k not in _FLAGS and _option(k) is None:
                raise ValueError(f"{k} is not a valid global option for theseus_amd.")
        _forward_to_reference(dict(options))
        return
    for k, v in options.items():
        if k in _FLAGS:
            _FLAGS[k] = bool(v)
            continue
        hit = _option(k)
        if hit is None:
            raise ValueError(f"{k} is not a valid global option for theseus_amd (this path reads the so3_* / se2_* "
                             f"near_zero / d_near_zero / near_pi thresholds and fast_approx_local_jacobians).")
        table, dt, attr = hit
        table[dt][attr] = float(v)


def reset_global_params() -> None:
    if _REFERENCE_PARAMS is not None:
        import torchlie
        import theseus
        torchlie.reset_global_params()
        theseus.set_global_params(dict(fast_approx_local_jacobians=False, se2_near_zero_eps_float32=3e-2,
                                       se2_d_near_zero_eps_float32=1e-1, se2_near_zero_eps_float64=1e-6,
                                       se2_d_near_zero_eps_float64=1e-3))
    _LIE_EPS[torch.float32].update(near_zero=1e-2, d_near_zero=2e-1, near_pi=1e-2)
    _LIE_EPS[torch.float64].update(near_zero=5e-3, d_near_zero=1e-2, near_pi=1e-7)
    _SE2_EPS[torch.float32].update(near_zero=3e-2, d_near_zero=1e-1)
    _SE2_EPS[torch.float64].update(near_zero=1e-6, d_near_zero=1e-3)
    _FLAGS["fast_approx_local_jacobians"] = False


def set_lie_eps(dtype, **kw):
    """Short form of set_global_params for the three so3 thresholds."""
    tag = "float32" if dtype == torch.float32 else "float64"
    for k, v in kw.items():
        if k not in _LIE_EPS[dtype]:
            raise KeyError(k)
        if _REFERENCE_PARAMS is not None:
            _forward_to_reference({f"so3_{k}_eps_{tag}": float(v)})
        _LIE_EPS[dtype][k] = float(v)


def set_se2_eps(dtype, **kw):
    """Short form of set_global_params for se2_near_zero_eps / se2_d_near_zero_eps."""
    tag = "float32" if dtype == torch.float32 else "float64"
    for k, v in kw.items():
        if k not in _SE2_EPS[dtype]:
            raise KeyError(k)
        if _REFERENCE_PARAMS is not None:
            _forward_to_reference({f"se2_{k}_eps_{tag}": float(v)})
        _SE2_EPS[dtype][k] = float(v)


def lie_eps(dtype) -> _lib.LieEps:
    r = _REFERENCE_PARAMS
    if r is not None:
        return _lib.LieEps(r.get_eps("so3", "near_zero", dtype), r.get_eps("so3", "d_near_zero", dtype),
                           r.get_eps("so3", "near_pi", dtype))
    e = _LIE_EPS[dtype]
    return _lib.LieEps(e["near_zero"], e["d_near_zero"], e["near_pi"])


def se2_eps(dtype) -> _lib.SE2Eps:
    r = _REFERENCE_PARAMS
    if r is not None:
        return _lib.SE2Eps(r.get_eps("se2", "near_zero", dtype), r.get_eps("se2", "d_near_zero", dtype))
    e = _SE2_EPS[dtype]
    return _lib.SE2Eps(e["near_zero"], e["d_near_zero"])


def fast_approx_local_jacobians() -> bool:
    """theseus/embodied/misc/local_cost_fn.py:43-57: Local/Difference costs then use the identity as their Jacobian."""
    r = _REFERENCE_PARAMS
    return bool(r.fast_approx_local_jacobians) if r is not None else _FLAGS["fast_approx_local_jacobians"]


def round_up(x, m):
    return (x + m - 1) // m * m


GROUP_RECORD = {"SE3": (12, 6), "SO3": (9, 3), "SE2": (4, 3), "SO2": (2, 1)}   # (scalars per record, dof)


def group_of(poses: torch.Tensor) -> str:
    if poses.dim() == 3:
        return "SO2" if poses.shape[-1] == 2 else "SE2"
    return "SO3" if poses.shape[-1] == 3 else "SE3"


@dataclass
class PGTensors:
    """Per-call tensors of a pose-graph objective in the entity-major device layout."""

    poses: torch.Tensor          # (P, B, 3, 4)     SE2: (P, B, 4)
    meas: torch.Tensor           # (E, Bm, 3, 4)   Bm in {1, B}
    w_between: torch.Tensor      # (E, Bw, 6)       SE2: (E, Bw, 3)
    prior_target: torch.Tensor   # (K, Bt, 3, 4)
    w_prior: torch.Tensor        # (K, Bw, 6)
    # RobustCostFunction wrappers: loss code (_lib.LOSS_* [| _lib.LOSS_FLATTEN]) per role and log_loss_radius (E|K, Br, 1);
    # loss_<role>: (E|K,) int32 code per cost when the costs of one role differ (robust_<role> is then any non-zero code)
    robust_between: int = 0
    log_radius_between: Optional[torch.Tensor] = None
    robust_prior: int = 0
    log_radius_prior: Optional[torch.Tensor] = None
    loss_between: Optional[torch.Tensor] = None
    loss_prior: Optional[torch.Tensor] = None

    def without_robust(self) -> "PGTensors":
        """The same costs with their RobustCostFunction wrappers taken off."""
        return dataclasses.replace(self, robust_between=0, robust_prior=0, loss_between=None, loss_prior=None)

    @property
    def batch(self):
        return self.poses.shape[1]

    @property
    def se2(self) -> bool:
        return self.poses.dim() == 3 and self.poses.shape[-1] == 4

    @property
    def group(self) -> str:
        """Read off the record shape: SE3 (3,4), SO3 (3,3), SE2 (4,), SO2 (2,)."""
        return group_of(self.poses)

    def c_struct(self, poses: Optional[torch.Tensor] = None) -> _lib.PGData:
        poses = self.poses if poses is None else poses
        B = poses.shape[1]
        d = _lib.PGData()
        d.batch = B
        d.poses = _lib.ptr(poses, "poses").value

        def put(name, t, width):
            nb = t.shape[1]
            if nb not in (1, B):
                raise ValueError(f"{name}: batch dimension {nb} is neither 1 nor {B}")
            setattr(d, name, _lib.ptr(t, name).value if t.numel() else None)
            setattr(d, name + "_bstride", width if nb == B else 0)

        gw, dof = GROUP_RECORD[group_of(poses)]
        put("meas", self.meas, gw)
        put("w_between", self.w_between, dof)
        put("prior_target", self.prior_target, gw)
        put("w_prior", self.w_prior, dof)
        for role in ("between", "prior"):
            kind = getattr(self, "robust_" + role)
            setattr(d, "robust_" + role, int(kind))
            table = getattr(self, "loss_" + role)
            if kind:
                put("log_radius_" + role, getattr(self, "log_radius_" + role), 1)
                if table is not None:
                    if table.dtype != torch.int32 or table.shape != (getattr(self, "log_radius_" + role).shape[0],):
                        raise ValueError(f"loss_{role}: one int32 loss code per cost")
                    setattr(d, "loss_" + role, _lib.ptr(table, "loss_" + role).value)
            elif table is not None:
                raise ValueError(f"loss_{role}: a per-cost loss table needs a non-zero robust_{role}")
        return d


class HipKernels:
    """Thin, allocation-free wrappers: every output buffer is supplied by the caller."""

    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        # per-call schedule of the factorisations (include/theseus_hip.h: thx_chol_schedule), handed to every thx_chol_factor* call
        # of THIS kernels object: -1 = the library default.  The library itself keeps no schedule state.
        self.chol_schedule = _lib.CholSchedule(-1, -1, -1, -1, -1, -1)

    def _sched(self):
        import ctypes
        return ctypes.byref(self.chol_schedule)

    # ---- SE3 elementwise ----------------------------------------------------------------------
    def se3_exp(self, xi, jac=False):
        xi = xi.contiguous()
        N = xi.shape[0]
        X = xi.new_empty(N, 3, 4)
        J = xi.new_empty(N, 6, 6) if jac else None
        e = lie_eps(xi.dtype)
        _lib.check(self.lib.thx_se3_exp(_lib.ptr(xi), _lib.ptr(X), _lib.ptr(J), N, _lib.dtype_code(xi.dtype), e,
                                        _lib.stream_ptr(xi.device)), "thx_se3_exp")
        return (X, J) if jac else X

    def se3_log(self, X, jac=False):
        X = X.contiguous()
        N = X.shape[0]
        xi = X.new_empty(N, 6)
        J = X.new_empty(N, 6, 6) if jac else None
        e = lie_eps(X.dtype)
        _lib.check(self.lib.thx_se3_log(_lib.ptr(X), _lib.ptr(xi), _lib.ptr(J), N, _lib.dtype_code(X.dtype), e,
                                        _lib.stream_ptr(X.device)), "thx_se3_log")
        return (xi, J) if jac else xi

    def se3_compose(self, X, Y):
        X, Y = X.contiguous(), Y.contiguous()
        Z = torch.empty_like(X)
        _lib.check(self.lib.thx_se3_compose(_lib.ptr(X), _lib.ptr(Y), _lib.ptr(Z), X.shape[0],
                                            _lib.dtype_code(X.dtype), _lib.stream_ptr(X.device)), "thx_se3_compose")
        return Z

    def se3_inverse(self, X):
        X = X.contiguous()
        Y = torch.empty_like(X)
        _lib.check(self.lib.thx_se3_inverse(_lib.ptr(X), _lib.ptr(Y), X.shape[0], _lib.dtype_code(X.dtype),
                                            _lib.stream_ptr(X.device)), "thx_se3_inverse")
        return Y

    def se3_adjoint(self, X):
        X = X.contiguous()
        A = X.new_empty(X.shape[0], 6, 6)
        _lib.check(self.lib.thx_se3_adjoint(_lib.ptr(X), _lib.ptr(A), X.shape[0], _lib.dtype_code(X.dtype),
                                            _lib.stream_ptr(X.device)), "thx_se3_adjoint")
        return A

    # ---- SE2 elementwise (theseus/geometry/se2.py) ---------------------------------------------
    def _se2_op(self, op, a, b, out, jac):
        N = a.shape[0]
        _lib.check(self.lib.thx_se2_op(op, _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.ptr(jac), N,
                                       _lib.dtype_code(a.dtype), se2_eps(a.dtype), _lib.stream_ptr(a.device)),
                   "thx_se2_op")

    def se2_exp(self, xi, jac=False):
        xi = xi.contiguous()
        X = xi.new_empty(xi.shape[0], 4)
        J = xi.new_empty(xi.shape[0], 3, 3) if jac else None
        self._se2_op(0, xi, None, X, J)
        return (X, J) if jac else X

    def se2_log(self, X, jac=False):
        X = X.contiguous()
        xi = X.new_empty(X.shape[0], 3)
        J = X.new_empty(X.shape[0], 3, 3) if jac else None
        self._se2_op(1, X, None, xi, J)
        return (xi, J) if jac else xi

    def se2_compose(self, X, Y):
        X, Y = X.contiguous(), Y.contiguous()
        Z = torch.empty_like(X)
        self._se2_op(2, X, Y, Z, None)
        return Z

    def se2_inverse(self, X):
        X = X.contiguous()
        Y = torch.empty_like(X)
        self._se2_op(3, X, None, Y, None)
        return Y

    def se2_adjoint(self, X):
        X = X.contiguous()
        A = X.new_empty(X.shape[0], 3, 3)
        self._se2_op(4, X, None, A, None)
        return A

    # ---- SO3 elementwise (theseus/geometry/so3.py over torchlie's so3_impl.py) -------------------
    def _so3_op(self, op, a, b, out, jac):
        _lib.check(self.lib.thx_so3_op(op, _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.ptr(jac), a.shape[0],
                                       _lib.dtype_code(a.dtype), lie_eps(a.dtype), _lib.stream_ptr(a.device)),
                   "thx_so3_op")

    def so3_exp(self, w, jac=False):
        w = w.contiguous()
        R = w.new_empty(w.shape[0], 3, 3)
        J = w.new_empty(w.shape[0], 3, 3) if jac else None
        self._so3_op(0, w, None, R, J)
        return (R, J) if jac else R

    def so3_log(self, R, jac=False):
        R = R.contiguous()
        w = R.new_empty(R.shape[0], 3)
        J = R.new_empty(R.shape[0], 3, 3) if jac else None
        self._so3_op(1, R, None, w, J)
        return (w, J) if jac else w

    def so3_compose(self, X, Y):
        X, Y = X.contiguous(), Y.contiguous()
        Z = torch.empty_like(X)
        self._so3_op(2, X, Y, Z, None)
        return Z

    def so3_inverse(self, X):
        X = X.contiguous()
        Y = torch.empty_like(X)
        self._so3_op(3, X, None, Y, None)
        return Y

    def so3_adjoint(self, X):
        X = X.contiguous()
        A = torch.empty_like(X)
        self._so3_op(4, X, None, A, None)
        return A

    # ---- SO2 elementwise (theseus/geometry/so2.py:167-235: records [cos, sin], tangent theta, every Jacobian 1) ----
    def _so2_op(self, op, a, b, out, jac):
        _lib.check(self.lib.thx_so2_op(op, _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.ptr(jac), a.shape[0],
                                       _lib.dtype_code(a.dtype), _lib.stream_ptr(a.device)), "thx_so2_op")

    def so2_exp(self, theta, jac=False):
        theta = theta.contiguous()
        X = theta.new_empty(theta.shape[0], 2)
        J = theta.new_empty(theta.shape[0], 1, 1) if jac else None
        self._so2_op(0, theta, None, X, J)
        return (X, J) if jac else X

    def so2_log(self, X, jac=False):
        X = X.contiguous()
        th = X.new_empty(X.shape[0], 1)
        J = X.new_empty(X.shape[0], 1, 1) if jac else None
        self._so2_op(1, X, None, th, J)
        return (th, J) if jac else th

    def so2_compose(self, X, Y):
        X, Y = X.contiguous(), Y.contiguous()
        Z = torch.empty_like(X)
        self._so2_op(2, X, Y, Z, None)
        return Z

    def so2_inverse(self, X):
        X = X.contiguous()
        Y = torch.empty_like(X)
        self._so2_op(3, X, None, Y, None)
        return Y

    def so2_adjoint(self, X):
        X = X.contiguous()
        A = X.new_empty(X.shape[0], 1, 1)
        self._so2_op(4, X, None, A, None)
        return A

    # ---- pose graph (SE3 records (3,4) -> thx_pg_*, SE2 records (4,) -> thx_pg2_*, SO3 records (3,3) -> thx_pgso3_*, SO2 records
    #      (2,) -> thx_pgso2_*) ----
    def pg_assemble(self, s: DeviceStructure, t: PGTensors, H, g, poses=None):
        d = t.c_struct(poses)
        dt = H.dtype
        if t.group == "SO2":
            _lib.check(self.lib.thx_pgso2_assemble(s.c, d, _lib.ptr(H, "H"), H.shape[-1], _lib.ptr(g, "g"),
                                                   _lib.dtype_code(dt), _lib.stream_ptr(H.device)), "thx_pgso2_assemble")
            return
        if t.group == "SO3":
            _lib.check(self.lib.thx_pgso3_assemble(s.c, d, _lib.ptr(H, "H"), H.shape[-1], _lib.ptr(g, "g"),
                                                   _lib.dtype_code(dt), lie_eps(dt), _lib.stream_ptr(H.device)),
                       "thx_pgso3_assemble")
            return
        if t.se2:
            _lib.check(self.lib.thx_pg2_assemble(s.c, d, _lib.ptr(H, "H"), H.shape[-1], _lib.ptr(g, "g"),
                                                 _lib.dtype_code(dt), se2_eps(dt), _lib.stream_ptr(H.device)),
                       "thx_pg2_assemble")
            return
        _lib.check(self.lib.thx_pg_assemble(s.c, d, _lib.ptr(H, "H"), H.shape[-1], _lib.ptr(g, "g"),
                                            _lib.dtype_code(dt), lie_eps(dt), _lib.stream_ptr(H.device)),
                   "thx_pg_assemble")

    # ---- block-compact Hessian (include/theseus_hip.h: thx_hblock_layout; theseus_amd/compiler.py:HessianBlocks) ----
    def pg_assemble_blocks(self, s: DeviceStructure, t: PGTensors, hb, Hc, g, poses=None):
        """thx_pg_assemble writing the block list ``Hc`` (B, bstride) instead of the dense frame (SE3 pose graphs)."""
        if t.group != "SE3":
            raise RuntimeError("the block-compact Hessian is implemented for SE3 pose graphs")
        dt = Hc.dtype
        _lib.check(self.lib.thx_pg_assemble_blocks(s.c, t.c_struct(poses), hb.c, _lib.ptr(Hc, "Hc"), Hc.stride(0), _lib.ptr(g, "g"),
                                                   _lib.dtype_code(dt), lie_eps(dt), _lib.stream_ptr(Hc.device)),
                   "thx_pg_assemble_blocks")

    def hblocks_expand(self, hb, Hc, H):
        """Block list -> the dense (B, ld, ld) frame (H must be zero where no block lies)."""
        _lib.check(self.lib.thx_hblocks_expand(hb.c, _lib.ptr(Hc), Hc.stride(0), Hc.shape[0], _lib.ptr(H), H.shape[-1],
                                               _lib.dtype_code(Hc.dtype), _lib.stream_ptr(Hc.device)), "thx_hblocks_expand")

    def hblocks_diag(self, hb, Hc, d):
        _lib.check(self.lib.thx_hblocks_diag(hb.c, _lib.ptr(Hc), Hc.stride(0), Hc.shape[0], _lib.ptr(d), d.stride(0),
                                             _lib.dtype_code(Hc.dtype), _lib.stream_ptr(Hc.device)), "thx_hblocks_diag")

    def chol_factor_hblocks(self, hb, Hc, n, damping, ellipsoidal, damping_eps, L, panels, info, pattern=None, rhs=None, y=None):
        """thx_chol_factor_forward / thx_chol_factor_sparse (``pattern``) with H read from the block list.  ``L`` (B, ld, ld): the
        dense frame; (B, nslots, 128, 128): the TILE-PACKED factor of ``pattern`` (ld = 0 in the C ABI)."""
        B, ld = L.shape[0], (0 if L.dim() == 4 else L.shape[-1])
        _lib.check(self.lib.thx_chol_factor_hblocks(
            hb.c, _lib.ptr(Hc), Hc.stride(0), n, B, _lib.ptr(damping), int(bool(ellipsoidal)), float(damping_eps), _lib.ptr(L), ld,
            _lib.ptr(panels), _lib.ptr(info), _lib.ptr(rhs), _lib.ptr(y), rhs.stride(0) if rhs is not None else n,
            pattern.c_struct(L.device) if pattern is not None else None, _lib.dtype_code(L.dtype), _lib.stream_ptr(L.device),
            self._sched()), "thx_chol_factor_hblocks")

    # ---- level-scheduled tile-sparse Cholesky (include/theseus_hip.h: thx_level_schedule; theseus_amd/sparse.py:LevelPattern) ----
    def chol_factor_levels(self, layout, Hc, damping, ellipsoidal, damping_eps, L, panels, info, pattern, rhs=None, y=None):
        """thx_chol_factor_levels: ``layout`` = the block list's piece tables for the pattern's PADDED tiles; L tile-packed;
        rhs / y: padded vectors of the fused forward substitution."""
        dev = L.device
        _lib.check(self.lib.thx_chol_factor_levels(
            layout.c, _lib.ptr(Hc), Hc.stride(0), L.shape[0], _lib.ptr(damping), int(bool(ellipsoidal)), float(damping_eps),
            _lib.ptr(L), _lib.ptr(panels), _lib.ptr(info), _lib.ptr(rhs), _lib.ptr(y), y.stride(0) if y is not None else 0,
            pattern.c_struct(dev), pattern.c_levels(dev), _lib.dtype_code(L.dtype), _lib.stream_ptr(dev), self._sched()),
            "thx_chol_factor_levels")

    def chol_solve_levels(self, L, panels, rhs, x, pattern, which=0):
        """thx_chol_solve_levels on vectors of the padded order (which: 0 both, 1 backward only, 2 forward only)."""
        dev = L.device
        _lib.check(self.lib.thx_chol_solve_levels(
            _lib.ptr(L), L.shape[0], _lib.ptr(panels), _lib.ptr(rhs), _lib.ptr(x), x.stride(0), int(which), pattern.c_struct(dev),
            pattern.c_solve_levels(dev), _lib.dtype_code(L.dtype), _lib.stream_ptr(dev)), "thx_chol_solve_levels")

    def vec_gather(self, src, dst, idx):
        """dst[b, k] = src[b, idx[k]] (0 where idx[k] < 0)."""
        if dst.dim() != 2 or src.dim() != 2 or dst.shape[1] != idx.numel() or dst.shape[0] != src.shape[0] or idx.dtype != torch.int32:
            raise ValueError(f"vec_gather: dst {tuple(dst.shape)} / src {tuple(src.shape)} / idx {tuple(idx.shape)} {idx.dtype} do not fit")
        _lib.check(self.lib.thx_vec_gather(_lib.ptr(src), src.stride(0), _lib.ptr(dst), dst.stride(0), _lib.ptr(idx), dst.shape[1],
                                           dst.shape[0], _lib.dtype_code(src.dtype), _lib.stream_ptr(src.device)), "thx_vec_gather")

    def pg_error(self, s: DeviceStructure, t: PGTensors, partials, err, poses=None):
        d = t.c_struct(poses)
        dt = err.dtype
        if t.group == "SO2":
            _lib.check(self.lib.thx_pgso2_error(s.c, d, _lib.ptr(partials), _lib.ptr(err), _lib.dtype_code(dt),
                                                _lib.stream_ptr(err.device)), "thx_pgso2_error")
            return
        if t.group == "SO3":
            _lib.check(self.lib.thx_pgso3_error(s.c, d, _lib.ptr(partials), _lib.ptr(err), _lib.dtype_code(dt),
                                                lie_eps(dt), _lib.stream_ptr(err.device)), "thx_pgso3_error")
            return
        if t.se2:
            _lib.check(self.lib.thx_pg2_error(s.c, d, _lib.ptr(partials), _lib.ptr(err), _lib.dtype_code(dt),
                                              se2_eps(dt), _lib.stream_ptr(err.device)), "thx_pg2_error")
            return
        _lib.check(self.lib.thx_pg_error(s.c, d, _lib.ptr(partials), _lib.ptr(err), _lib.dtype_code(dt), lie_eps(dt),
                                         _lib.stream_ptr(err.device)), "thx_pg_error")

    def pg_jacobians(self, s: DeviceStructure, t: PGTensors, J0, J1, eb, Jp, ep, poses=None):
        d = t.c_struct(poses)
        dt = t.poses.dtype
        if t.group == "SO2":
            _lib.check(self.lib.thx_pgso2_jacobians(s.c, d, _lib.ptr(J0), _lib.ptr(J1), _lib.ptr(eb), _lib.ptr(Jp), _lib.ptr(ep),
                                                    _lib.dtype_code(dt), _lib.stream_ptr(t.poses.device)), "thx_pgso2_jacobians")
            return
        if t.group == "SO3":
            _lib.check(self.lib.thx_pgso3_jacobians(s.c, d, _lib.ptr(J0), _lib.ptr(J1), _lib.ptr(eb), _lib.ptr(Jp),
                                                    _lib.ptr(ep), _lib.dtype_code(dt), lie_eps(dt),
                                                    _lib.stream_ptr(t.poses.device)), "thx_pgso3_jacobians")
            return
        if t.se2:
            _lib.check(self.lib.thx_pg2_jacobians(s.c, d, _lib.ptr(J0), _lib.ptr(J1), _lib.ptr(eb), _lib.ptr(Jp),
                                                  _lib.ptr(ep), _lib.dtype_code(dt), se2_eps(dt),
                                                  _lib.stream_ptr(t.poses.device)), "thx_pg2_jacobians")
            return
        _lib.check(self.lib.thx_pg_jacobians(s.c, d, _lib.ptr(J0), _lib.ptr(J1), _lib.ptr(eb), _lib.ptr(Jp),
                                             _lib.ptr(ep), _lib.dtype_code(dt), lie_eps(dt),
                                             _lib.stream_ptr(t.poses.device)), "thx_pg_jacobians")

    def retract(self, poses, delta, step, ignore_mask, out):
        """X <- X exp(step * delta) on the packed pose buffer; the group is read off the record shape."""
        grp = group_of(poses)
        if grp == "SE3":
            return self.se3_retract(poses, delta, step, ignore_mask, out)
        P, B = poses.shape[:2]
        dt = poses.dtype
        if grp == "SO2":
            _lib.check(self.lib.thx_so2_retract(_lib.ptr(poses), _lib.ptr(delta), delta.stride(0), float(step),
                                                _lib.ptr(ignore_mask), _lib.ptr(out), P, B, _lib.dtype_code(dt),
                                                _lib.stream_ptr(poses.device)), "thx_so2_retract")
            return
        if grp == "SO3":
            _lib.check(self.lib.thx_so3_retract(_lib.ptr(poses), _lib.ptr(delta), delta.stride(0), float(step),
                                                _lib.ptr(ignore_mask), _lib.ptr(out), P, B, _lib.dtype_code(dt),
                                                lie_eps(dt), _lib.stream_ptr(poses.device)), "thx_so3_retract")
            return
        _lib.check(self.lib.thx_se2_retract(_lib.ptr(poses), _lib.ptr(delta), delta.stride(0), float(step),
                                            _lib.ptr(ignore_mask), _lib.ptr(out), P, B, _lib.dtype_code(dt),
                                            se2_eps(dt), _lib.stream_ptr(poses.device)), "thx_se2_retract")

    def se3_retract(self, poses, delta, step, ignore_mask, out):
        P, B = poses.shape[:2]
        dt = poses.dtype
        _lib.check(self.lib.thx_se3_retract(_lib.ptr(poses), _lib.ptr(delta), delta.stride(0), float(step),
                                            _lib.ptr(ignore_mask), _lib.ptr(out), P, B, _lib.dtype_code(dt),
                                            lie_eps(dt), _lib.stream_ptr(poses.device)), "thx_se3_retract")

    # ---- bundle adjustment (csrc/ba_kernels.hip) --------------------------------------------------
    def ba_assemble(self, s, t, Hcc, Hpp, W, gd, g, diag):
        """Block quantities Hcc, Hpp, W, gd are fp64 buffers for every dtype (include/theseus_hip.h, PRECISION)."""
        d = t.c_struct()
        dt = g.dtype
        for b in (Hcc, Hpp, W, gd):
            assert b.dtype == torch.float64
        _lib.check(self.lib.thx_ba_assemble(s.c, d, _lib.ptr(Hcc), _lib.ptr(Hpp), _lib.ptr(W), _lib.ptr(gd), _lib.ptr(g),
                                            _lib.ptr(diag), g.stride(0), _lib.dtype_code(dt), lie_eps(dt),
                                            _lib.stream_ptr(g.device)), "thx_ba_assemble")

    def ba_schur(self, s, Hcc, Hpp, W, gd, damping, ellipsoidal, damping_eps, S, rhs, Hinv, tvec, info):
        B = gd.shape[0]
        _lib.check(self.lib.thx_ba_schur(s.c, B, _lib.ptr(Hcc), _lib.ptr(Hpp), _lib.ptr(W), _lib.ptr(gd), gd.stride(0),
                                         _lib.ptr(damping), int(bool(ellipsoidal)), float(damping_eps), _lib.ptr(S),
                                         S.shape[-1], _lib.ptr(rhs), rhs.stride(0), _lib.ptr(Hinv), _lib.ptr(tvec),
                                         _lib.ptr(info), _lib.dtype_code(S.dtype), _lib.stream_ptr(S.device)), "thx_ba_schur")

    def ba_schur_blocks(self, s, Hcc, Hpp, W, gd, damping, ellipsoidal, damping_eps, Sc, diag_blk, blk_dst, rhs, Hinv, tvec, info):
        """thx_ba_schur_blocks: S as a block list ``Sc`` (B, bstride), 36 contiguous values per 6 x 6 block (diag_blk / blk_dst:
        device int32 block ids, bit 30 of blk_dst = stored transposed)."""
        B = gd.shape[0]
        _lib.check(self.lib.thx_ba_schur_blocks(s.c, B, _lib.ptr(Hcc), _lib.ptr(Hpp), _lib.ptr(W), _lib.ptr(gd), gd.stride(0),
                                                _lib.ptr(damping), int(bool(ellipsoidal)), float(damping_eps), _lib.ptr(Sc),
                                                Sc.stride(0), _lib.ptr(diag_blk), _lib.ptr(blk_dst), _lib.ptr(rhs), rhs.stride(0),
                                                _lib.ptr(Hinv), _lib.ptr(tvec), _lib.ptr(info), _lib.dtype_code(Sc.dtype),
                                                _lib.stream_ptr(Sc.device)), "thx_ba_schur_blocks")

    def ba_backsub(self, s, W, Hinv, tvec, delta):
        B = delta.shape[0]
        _lib.check(self.lib.thx_ba_backsub(s.c, B, _lib.ptr(W), _lib.ptr(Hinv), _lib.ptr(tvec), _lib.ptr(delta),
                                           delta.stride(0), _lib.dtype_code(delta.dtype), _lib.stream_ptr(delta.device)),
                   "thx_ba_backsub")

    def ba_error(self, s, t, partials, err, cams=None, points=None):
        d = t.c_struct(cams, points)
        dt = err.dtype
        _lib.check(self.lib.thx_ba_error(s.c, d, _lib.ptr(partials), _lib.ptr(err), _lib.dtype_code(dt), lie_eps(dt),
                                         _lib.stream_ptr(err.device)), "thx_ba_error")

    def ba_av(self, s, t, v, rows, out_t):
        """thx_ba_av: out_t (m, B) = (A v)^T; ``rows`` = (obs_row, cam_prior_row, pt_prior_row) device int32 tensors."""
        d = t.c_struct()
        dt = v.dtype
        _lib.check(self.lib.thx_ba_av(s.c, d, _lib.ptr(v), v.stride(0), _lib.ptr(rows[0]), _lib.ptr(rows[1]), _lib.ptr(rows[2]),
                                      _lib.ptr(out_t), _lib.dtype_code(dt), lie_eps(dt), _lib.stream_ptr(v.device)), "thx_ba_av")

    def ba_vjp(self, s, t, w, grads):
        """grads: dict name -> preallocated tensor | None for feat, w_obs, focal, k1, k2, log_radius, cam_prior_target,
        w_cam_prior, pt_prior_target, w_pt_prior (thx_ba_vjp's outputs, in that order)."""
        d = t.c_struct()
        dt = w.dtype
        order = ("feat", "w_obs", "focal", "k1", "k2", "log_radius", "cam_prior_target", "w_cam_prior", "pt_prior_target",
                 "w_pt_prior")
        _lib.check(self.lib.thx_ba_vjp(s.c, d, _lib.ptr(w), w.stride(0), *[_lib.ptr(grads.get(k)) for k in order],
                                       _lib.dtype_code(dt), lie_eps(dt), _lib.stream_ptr(w.device)), "thx_ba_vjp")

    def ba_unroll_vjp(self, s, t, w, delta, grads, ell_damping=None):
        """thx_ba_unroll_vjp: ``grads`` = dict name -> preallocated tensor over _lib.BA_UNROLL_GRADS (log_radius_obs may be
        missing); ``ell_damping`` (B,) lambda with ellipsoidal damping, else None."""
        d = t.c_struct()
        dt = w.dtype
        out = _lib.BAUnrollGrads()
        for k in _lib.BA_UNROLL_GRADS:
            g = grads.get(k)
            setattr(out, k, _lib.ptr(g).value if g is not None and g.numel() else None)
        _lib.check(self.lib.thx_ba_unroll_vjp(s.c, d, _lib.ptr(w), w.stride(0), _lib.ptr(delta), delta.stride(0),
                                              _lib.ptr(ell_damping), out, _lib.dtype_code(dt), lie_eps(dt),
                                              _lib.stream_ptr(w.device)), "thx_ba_unroll_vjp")

    def copy_where(self, mask, src, dst):
        """dst[k, b] <- src[k, b] where mask[b]; src / dst (N, B, ...) contiguous, mask (B,) bool or uint8."""
        if src.shape != dst.shape or src.dtype != dst.dtype or not (src.is_contiguous() and dst.is_contiguous()):
            raise ValueError("copy_where: src and dst must be contiguous tensors of one shape and dtype")
        N, B = dst.shape[0], dst.shape[1]
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        rec = dst[0, 0].numel() * dst.element_size()
        _lib.check(self.lib.thx_copy_where(_lib.ptr(mask), _lib.ptr(src), _lib.ptr(dst), N, B, rec,
                                           _lib.stream_ptr(dst.device)), "thx_copy_where")

    def vec_retract(self, x, delta, col0, step, ignore_mask, out):
        N, B, dof = x.shape
        _lib.check(self.lib.thx_vec_retract(_lib.ptr(x), _lib.ptr(delta), delta.stride(0), int(col0), float(step),
                                            _lib.ptr(ignore_mask), _lib.ptr(out), N, dof, B, _lib.dtype_code(x.dtype),
                                            _lib.stream_ptr(x.device)), "thx_vec_retract")

    def lm_accept_diag(self, delta, g, diag, n, damping, prev_err, new_err, ellipsoidal, accept, down, up, reject):
        B = delta.shape[0]
        _lib.check(self.lib.thx_lm_accept_diag(_lib.ptr(delta), _lib.ptr(g), _lib.ptr(diag), delta.stride(0), n, B,
                                               _lib.ptr(damping), _lib.ptr(prev_err), _lib.ptr(new_err),
                                               int(bool(ellipsoidal)), float(accept), float(down), float(up),
                                               _lib.ptr(reject), _lib.dtype_code(delta.dtype),
                                               _lib.stream_ptr(delta.device)), "thx_lm_accept_diag")

    # ---- generic block assembly ------------------------------------------------------------------
    def block_assemble(self, asm, jacobians, errors, H, g):
        """asm: theseus_amd.generic.BlockAssembler; jacobians[c][slot] (B|1, dim, dof), errors[c] (B|1, dim)."""
        import numpy as np
        for Js in jacobians:
            for J in Js:
                _lib.ptr(J, "jacobian block")  # device / contiguity check
        h_terms, g_terms = asm.term_tables(jacobians, errors)
        ref = H if H is not None else g
        dev = ref.device
        ht_d, he2t_d, gt_d, ge2t_d = asm._static(dev)
        up = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dev)  # noqa: E731
        hterm_d, gterm_d = up(h_terms), up(g_terms)
        B = ref.shape[0]
        nh = asm.n_h_elems if H is not None else 0
        ng = asm.n_g_elems if g is not None else 0
        _lib.check(self.lib.thx_block_assemble(
            _lib.ptr(ht_d), _lib.ptr(hterm_d), _lib.ptr(he2t_d), nh, _lib.ptr(gt_d), _lib.ptr(gterm_d), _lib.ptr(ge2t_d), ng,
            _lib.ptr(H), H.shape[-1] if H is not None else 0, _lib.ptr(g), g.stride(0) if g is not None else 0, B,
            _lib.dtype_code(ref.dtype), _lib.stream_ptr(dev)), "thx_block_assemble")
        # keep the uploaded tables alive until the stream has consumed them
        self._keepalive = (hterm_d, gterm_d)

    # ---- implicit backward ----------------------------------------------------------------------
    def retract_vjp(self, poses, delta, step, grad_out, grad_delta):
        """grad_X_new -> grad_delta of X exp(step * delta); the group is read off the record shape."""
        if group_of(poses) == "SO3":
            P, B = poses.shape[:2]
            dt = poses.dtype
            _lib.check(self.lib.thx_so3_retract_vjp(_lib.ptr(poses), _lib.ptr(delta), delta.stride(0), float(step),
                                                    _lib.ptr(grad_out), _lib.ptr(grad_delta), grad_delta.stride(0), P, B,
                                                    _lib.dtype_code(dt), lie_eps(dt), _lib.stream_ptr(poses.device)),
                       "thx_so3_retract_vjp")
            return
        if poses.dim() == 4:
            return self.se3_retract_vjp(poses, delta, step, grad_out, grad_delta)
        P, B = poses.shape[:2]
        dt = poses.dtype
        _lib.check(self.lib.thx_se2_retract_vjp(_lib.ptr(poses), _lib.ptr(delta), delta.stride(0), float(step),
                                                _lib.ptr(grad_out), _lib.ptr(grad_delta), grad_delta.stride(0), P, B,
                                                _lib.dtype_code(dt), se2_eps(dt), _lib.stream_ptr(poses.device)),
                   "thx_se2_retract_vjp")

    def se3_retract_vjp(self, poses, delta, step, grad_out, grad_delta):
        P, B = poses.shape[:2]
        dt = poses.dtype
        _lib.check(self.lib.thx_se3_retract_vjp(_lib.ptr(poses), _lib.ptr(delta), delta.stride(0), float(step),
                                                _lib.ptr(grad_out), _lib.ptr(grad_delta), grad_delta.stride(0), P, B,
                                                _lib.dtype_code(dt), lie_eps(dt), _lib.stream_ptr(poses.device)),
                   "thx_se3_retract_vjp")

    def pg_vjp(self, s: DeviceStructure, t: PGTensors, w, g_meas, g_wb, g_tgt, g_wp, poses=None, g_lrb=None, g_lrp=None):
        d = t.c_struct(poses)
        dt = w.dtype
        if t.group == "SO3":
            _lib.check(self.lib.thx_pgso3_vjp(s.c, d, _lib.ptr(w), w.stride(0), _lib.ptr(g_meas), _lib.ptr(g_wb),
                                              _lib.ptr(g_tgt), _lib.ptr(g_wp), _lib.ptr(g_lrb), _lib.ptr(g_lrp),
                                              _lib.dtype_code(dt), lie_eps(dt), _lib.stream_ptr(w.device)), "thx_pgso3_vjp")
            return
        if t.se2:
            _lib.check(self.lib.thx_pg2_vjp(s.c, d, _lib.ptr(w), w.stride(0), _lib.ptr(g_meas), _lib.ptr(g_wb),
                                            _lib.ptr(g_tgt), _lib.ptr(g_wp), _lib.ptr(g_lrb), _lib.ptr(g_lrp),
                                            _lib.dtype_code(dt), se2_eps(dt), _lib.stream_ptr(w.device)), "thx_pg2_vjp")
            return
        _lib.check(self.lib.thx_pg_vjp(s.c, d, _lib.ptr(w), w.stride(0), _lib.ptr(g_meas), _lib.ptr(g_wb),
                                       _lib.ptr(g_tgt), _lib.ptr(g_wp), _lib.ptr(g_lrb), _lib.ptr(g_lrp),
                                       _lib.dtype_code(dt), lie_eps(dt),
                                       _lib.stream_ptr(w.device)), "thx_pg_vjp")

    def pg_unroll_vjp(self, s: DeviceStructure, t: PGTensors, w, delta, g_pose_i, g_pose_j, g_meas, g_wb, g_pose_p, g_tgt, g_wp,
                      poses=None, ell_damping=None, g_lrb=None, g_lrp=None):
        """thx_pg_unroll_vjp / thx_pg2_unroll_vjp / thx_pgso3_unroll_vjp (include/theseus_hip.h): the per-cost backward of one
        differentiated iteration of an SE3 / SE2 / SO3 pose graph."""
        dt = w.dtype
        fn, eps = {"SE3": ("thx_pg_unroll_vjp", lie_eps), "SE2": ("thx_pg2_unroll_vjp", se2_eps),
                   "SO3": ("thx_pgso3_unroll_vjp", lie_eps)}[t.group]
        _lib.check(getattr(self.lib, fn)(s.c, t.c_struct(poses), _lib.ptr(w), w.stride(0), _lib.ptr(delta), delta.stride(0),
                                         _lib.ptr(ell_damping), _lib.ptr(g_pose_i), _lib.ptr(g_pose_j), _lib.ptr(g_meas), _lib.ptr(g_wb),
                                         _lib.ptr(g_pose_p), _lib.ptr(g_tgt), _lib.ptr(g_wp), _lib.ptr(g_lrb), _lib.ptr(g_lrp),
                                         _lib.dtype_code(dt), eps(dt), _lib.stream_ptr(w.device)), fn)

    # ---- dense solver ---------------------------------------------------------------------------
    def chol_factor(self, H, n, damping, ellipsoidal, damping_eps, L, panels, info, rhs=None, y=None):
        """L L^T = H + damping.  With rhs/y the forward substitution y = L^-1 rhs is fused in."""
        B, ld = H.shape[0], H.shape[-1]
        common = (_lib.ptr(H), ld, n, B, _lib.ptr(damping), int(bool(ellipsoidal)), float(damping_eps), _lib.ptr(L),
                  _lib.ptr(panels), _lib.ptr(info))
        if rhs is None:
            _lib.check(self.lib.thx_chol_factor(*common, _lib.dtype_code(H.dtype), _lib.stream_ptr(H.device), self._sched()),
                       "thx_chol_factor")
        else:
            _lib.check(self.lib.thx_chol_factor_forward(*common, _lib.ptr(rhs), _lib.ptr(y), rhs.stride(0),
                                                        _lib.dtype_code(H.dtype), _lib.stream_ptr(H.device), self._sched()),
                       "thx_chol_factor_forward")

    def chol_factor_sparse(self, H, n, damping, ellipsoidal, damping_eps, L, panels, info, pattern, rhs=None, y=None):
        """thx_chol_factor_sparse: ``pattern`` is a theseus_amd.sparse.TilePattern (device tables of the symbolic tile
        factorisation); otherwise as chol_factor."""
        B, ld = H.shape[0], H.shape[-1]
        _lib.check(self.lib.thx_chol_factor_sparse(_lib.ptr(H), ld, n, B, _lib.ptr(damping), int(bool(ellipsoidal)),
                                                   float(damping_eps), _lib.ptr(L), _lib.ptr(panels), _lib.ptr(info),
                                                   _lib.ptr(rhs), _lib.ptr(y), rhs.stride(0) if rhs is not None else 0,
                                                   pattern.c_struct(H.device), _lib.dtype_code(H.dtype),
                                                   _lib.stream_ptr(H.device), self._sched()), "thx_chol_factor_sparse")

    def chol_split_diag_min_batch(self, min_batch: int) -> int:
        """Schedule of THIS kernels object's factorisations (include/theseus_hip.h: thx_chol_schedule.split_diag_min_batch): from
        ``min_batch`` problems per launch on, the diagonal phase runs as SYRK kernel + one-wave-per-tile kernel (-1: the library
        default).  Returns the previous value.  Per call in the C ABI: the library keeps no schedule state."""
        prev = int(self.chol_schedule.split_diag_min_batch)
        self.chol_schedule.split_diag_min_batch = min(int(min_batch), 2 ** 31 - 1)
        return prev

    def chol_column_pairs(self, on) -> int:
        """Schedule of THIS kernels object's fp32 dense-frame factorisations (thx_chol_schedule.column_pairs): two block columns
        per off-diagonal launch, bit-identical factor (1 on, 0 off, -1 the library default).  Returns the previous setting."""
        prev = int(self.chol_schedule.column_pairs)
        self.chol_schedule.column_pairs = int(on)
        return prev

    def chol_right_looking_max_batch(self, max_batch: int) -> int:
        """Schedule of THIS kernels object's fp32 dense-frame factorisations (thx_chol_schedule.right_looking_max_batch): batches of
        at most ``max_batch`` problems take the right-looking schedule (0 never, -1 the library default: by dtype and size -- fp32 64 /
        fp64 40 problems up to 12 block columns, 32 from 24 on).  Returns the previous setting."""
        prev = int(self.chol_schedule.right_looking_max_batch)
        self.chol_schedule.right_looking_max_batch = int(max_batch)
        return prev

    def chol_hb_scatter_max_pieces(self, max_pieces: int) -> int:
        """Schedule of THIS kernels object's factorisations from a block-compact H (thx_chol_schedule.hb_scatter_max_pieces): layouts
        whose off-diagonal tiles hold at most ``max_pieces`` blocks have them added by the matrix cores, others gathered through LDS
        -- the same bits either way (0 always gather, -1 the library default).  Returns the previous setting."""
        prev = int(self.chol_schedule.hb_scatter_max_pieces)
        self.chol_schedule.hb_scatter_max_pieces = int(max_pieces)
        return prev

    def chol_f64_wide_max_ktiles(self, max_ktiles: int) -> int:
        """Schedule of THIS kernels object's fp64 factorisations (thx_chol_schedule.f64_wide_max_ktiles): the off-diagonal tiles of
        the first ``max_ktiles`` block columns come from eight-wave workgroups -- bit-identical to the four-wave kernel (0 never,
        -1 the library default).  Returns the previous setting."""
        prev = int(self.chol_schedule.f64_wide_max_ktiles)
        self.chol_schedule.f64_wide_max_ktiles = int(max_ktiles)
        return prev

    def chol_f64_half_max_ktiles(self, max_ktiles: int) -> int:
        """Schedule of THIS kernels object's fp64 factorisations (thx_chol_schedule.f64_half_max_ktiles): the off-diagonal tiles of
        the first ``max_ktiles`` block columns as half tiles from four-wave workgroups, four per CU -- bit-identical (0 never, -1
        the library default).  Returns the previous setting."""
        prev = int(self.chol_schedule.f64_half_max_ktiles)
        self.chol_schedule.f64_half_max_ktiles = int(max_ktiles)
        return prev

    def chol_solve(self, L, n, panels, rhs, x):
        B, ld = L.shape[0], L.shape[-1]
        _lib.check(self.lib.thx_chol_solve(_lib.ptr(L), ld, n, B, _lib.ptr(panels), _lib.ptr(rhs), _lib.ptr(x),
                                           rhs.stride(0), _lib.dtype_code(L.dtype), _lib.stream_ptr(L.device)),
                   "thx_chol_solve")

    def chol_solve_sparse(self, L, n, panels, rhs, x, pattern, backward_only=False):
        """thx_chol_solve_sparse: the triangular solves over the structurally non-zero tiles of L only (``pattern``: a
        theseus_amd.sparse.TilePattern); ``backward_only``: x = L^-T rhs.  ``L`` may be the tile-packed factor (4-D)."""
        B, ld = L.shape[0], (0 if L.dim() == 4 else L.shape[-1])
        _lib.check(self.lib.thx_chol_solve_sparse(_lib.ptr(L), ld, n, B, _lib.ptr(panels), _lib.ptr(rhs), _lib.ptr(x),
                                                  rhs.stride(0), int(bool(backward_only)), pattern.c_struct(L.device),
                                                  _lib.dtype_code(L.dtype), _lib.stream_ptr(L.device)), "thx_chol_solve_sparse")

    def chol_solve_backward(self, L, n, panels, y, x):
        B, ld = L.shape[0], L.shape[-1]
        _lib.check(self.lib.thx_chol_solve_backward(_lib.ptr(L), ld, n, B, _lib.ptr(panels), _lib.ptr(y), _lib.ptr(x),
                                                    y.stride(0), _lib.dtype_code(L.dtype), _lib.stream_ptr(L.device)),
                   "thx_chol_solve_backward")

    def diag(self, H, n, d):
        B, ld = H.shape[0], H.shape[-1]
        _lib.check(self.lib.thx_diag(_lib.ptr(H), ld, n, B, _lib.ptr(d), d.stride(0), _lib.dtype_code(H.dtype),
                                     _lib.stream_ptr(H.device)), "thx_diag")

    def lm_accept(self, delta, g, H, n, damping, prev_err, new_err, ellipsoidal, accept, down, up, reject):
        B = delta.shape[0]
        ld = H.shape[-1] if H is not None else 0
        _lib.check(self.lib.thx_lm_accept(_lib.ptr(delta), _lib.ptr(g), delta.stride(0), _lib.ptr(H), ld, n, B,
                                          _lib.ptr(damping), _lib.ptr(prev_err), _lib.ptr(new_err),
                                          int(bool(ellipsoidal)), float(accept), float(down), float(up),
                                          _lib.ptr(reject), _lib.dtype_code(delta.dtype),
                                          _lib.stream_ptr(delta.device)), "thx_lm_accept")


_default = None


def default_kernels() -> HipKernels:
    global _default
    if _default is None:
        _default = HipKernels()
    return _default
