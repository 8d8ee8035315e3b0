"""`LinearSolver` plugin surface + the HIP dense Cholesky solver.

ABC contract: theseus/optimizer/linear/linear_solver.py:15-37.  ``HipCholeskySolver`` replaces
``CholeskyDenseSolver`` (theseus/optimizer/linear/dense_solver.py:20-161): damping semantics of
``DenseSolver._apply_damping`` (:38-64, out of place), ``torch.linalg.cholesky`` +
``torch.cholesky_solve`` (:159-161) become one batched tiled HIP factorisation + two triangular
solves; a non positive-definite system raises ``RuntimeError`` like ``torch.linalg.cholesky`` does.
"""
import abc
import warnings
from typing import Any, Dict, Optional, Type, Union

import torch

from .core import Objective
from .linearization import HipLinearization, Linearization
from . import _lib


class LinearSolver(abc.ABC):
    def __init__(self, objective: Objective, linearization_cls: Optional[Type[Linearization]] = None,
                 linearization_kwargs: Optional[Dict[str, Any]] = None, **kwargs):
        linearization_kwargs = linearization_kwargs or {}
        self.linearization: Linearization = linearization_cls(objective, **linearization_kwargs)

    def reset(self, **kwargs):
        pass

    @abc.abstractmethod
    def solve(self, damping: Optional[Union[float, torch.Tensor]] = None, **kwargs) -> torch.Tensor:
        pass


class HipCholeskyCore:
    """Back-end half of the solver (buffers, factor, solves), shared by theseus_amd's mirror class below and
    the adapter for the real ``theseus`` (theseus_amd/plugin.py)."""

    def _core_init(self):
        self.K = self.linearization.K
        self.L = self.panels = self.info = self._y = None
        self._lam = None
        self.factor_version = 0  # bumped by every factorisation (the implicit backward checks its factor is current)

    def _ensure_buffers(self):
        lin = self.linearization
        g = lin.g     # (B, n): shape / dtype / device of the system (lin.H may be block-compact: never touched here)
        shape = (g.shape[0], lin.ld, lin.ld)
        if self.L is None or tuple(self.L.shape) != shape or self.L.device != g.device or self.L.dtype != g.dtype:
            B = g.shape[0]
            nt = (lin.n + _lib.THX_TILE - 1) // _lib.THX_TILE
            self.L = torch.zeros(shape, dtype=g.dtype, device=g.device)
            self.panels = torch.empty(B, nt, _lib.THX_TILE, _lib.THX_TILE, dtype=g.dtype, device=g.device)
            self._y = torch.empty(B, lin.n, dtype=g.dtype, device=g.device)
            self.info = torch.zeros(B, dtype=torch.int32, device=g.device)
            self._lam = torch.empty(B, dtype=g.dtype, device=g.device)
            self._unfused_forward = False   # (decided per system size: see _solve)

    def _linearized(self) -> bool:
        lin = self.linearization
        return lin.linearized if hasattr(lin, "linearized") else lin.H is not None

    def _factor_call(self, lam, ellipsoidal_damping, damping_eps, rhs, y, pattern=None):
        """One factorisation launch sequence: H from the block list when the linearization keeps it compact."""
        lin = self.linearization
        if getattr(lin, "_compact", False):
            self.K.chol_factor_hblocks(lin.hblocks, lin.Hc, lin.n, lam, ellipsoidal_damping, damping_eps, self.L, self.panels,
                                       self.info, pattern=pattern, rhs=rhs, y=y)
        elif pattern is not None:
            self.K.chol_factor_sparse(lin.H, lin.n, lam, ellipsoidal_damping, damping_eps, self.L, self.panels, self.info,
                                      pattern, rhs=rhs, y=y)
        else:
            self.K.chol_factor(lin.H, lin.n, lam, ellipsoidal_damping, damping_eps, self.L, self.panels, self.info, rhs=rhs, y=y)

    def factorize(self, damping: Optional[Union[float, torch.Tensor]] = None, ellipsoidal_damping: bool = True,
                  damping_eps: float = 1e-8, rhs: Optional[torch.Tensor] = None):
        """L L^T = AtA (+ damping); keeps L and the solve panels for later solves (the implicit backward
        re-uses them).  With ``rhs`` the forward substitution L y = rhs is fused into the factorisation
        (no extra pass over L) and y is returned."""
        if not self._linearized():
            raise RuntimeError("linearize() must be called before solve().")
        self._ensure_buffers()
        lam = None
        if damping is not None:
            lam = self._lam
            if isinstance(damping, torch.Tensor):
                lam.copy_(damping.to(lam.dtype).expand(lam.shape[0]))
            else:
                lam.fill_(float(damping))
        y = self._y if rhs is not None else None
        self.factor_version += 1
        self._factored_with = (lam is not None, bool(ellipsoidal_damping), float(damping_eps))   # (what L L^T is the factor of)
        self._factor_call(lam, ellipsoidal_damping, damping_eps, rhs, y)
        return y

    def solve_with_factor(self, rhs: torch.Tensor) -> torch.Tensor:
        """(L L^T)^-1 rhs with the cached factor (forward + backward substitution kernels)."""
        rhs = rhs.contiguous()
        x = torch.empty_like(rhs)
        self._substitute(rhs, x, backward_only=False)
        return x

    def factor_snapshot(self):
        """A private copy of the CURRENT factor, for a solve after the solver has factorised again (the differentiated iterations
        of BackwardMode.UNROLL / TRUNCATED: every iteration's backward solves with that iteration's factor)."""
        return self.L.clone(), self.panels.clone()

    def solve_with_snapshot(self, snapshot, rhs: torch.Tensor) -> torch.Tensor:
        """(L L^T)^-1 rhs with a factor kept by ``factor_snapshot`` (dense frame: thx_chol_solve)."""
        L, panels = snapshot
        rhs = rhs.contiguous()
        x = torch.empty_like(rhs)
        self.K.chol_solve(L, self.linearization.n, panels, rhs, x)
        return x

    def _substitute(self, rhs, x, backward_only: bool):
        """x = L^-T rhs (``backward_only``) or (L L^T)^-1 rhs with the current factor (dense frame: every tile of L)."""
        if backward_only:
            self.K.chol_solve_backward(self.L, self.linearization.n, self.panels, rhs, x)
        else:
            self.K.chol_solve(self.L, self.linearization.n, self.panels, rhs, x)

    def check_info(self):
        bad = self.info.nonzero()
        if bad.numel():
            b = int(bad[0])
            raise RuntimeError(
                f"linalg.cholesky: (Batch element {b}): The factorization could not be completed because the "
                f"input is not positive-definite (the leading minor of order {int(self.info[b])} is not "
                "positive-definite).")

    def singular_mask(self) -> torch.Tensor:
        """(B,) bool -- ``check_singular=True`` (dense_solver.py:91-103): the reference runs ``torch.lu`` on the UNDAMPED
        ``AtA`` and drops the batch items whose factorisation hits an exactly zero pivot.  For ``AtA = A^T A`` that is the
        case of an all-zero column of ``A`` (a variable every cost of which has zero weight): a zero on the diagonal of
        ``AtA``.  Rank deficiency that leaves a non-zero rounding residue (a gauge freedom) passes the reference's LU test
        too, and is solved here as it is there."""
        return (self.linearization.diagonal() == 0).any(dim=1)

    def dropped_mask(self) -> Optional[torch.Tensor]:
        """(B,) bool of the batch items the LAST ``_solve`` dropped under ``check_singular=True`` (zero step, dense_solver.py:91-103),
        else None.  The differentiated nodes keep it: a dropped item's step does not depend on anything, its gradients are zero
        (the reference's masked assignment), and its -- possibly broken -- factor must not reach the backward."""
        return getattr(self, "_last_singular", None) if getattr(self, "_check_singular", False) else None

    def post_singular_warning(self):
        """The reference's warning for ``check_singular=True`` (dense_solver.py:97-102) -- one host look at a device flag: called
        where the host synchronises anyway."""
        seen = getattr(self, "_singular_seen", None)
        self._singular_seen = None
        if seen is not None and bool(seen):
            warnings.warn("Singular matrix found in batch, solution will be set to all 0 for all singular matrices.", RuntimeWarning)

    def _solve(self, damping, ellipsoidal_damping, damping_eps, check_info) -> torch.Tensor:
        if damping is not None and isinstance(damping, torch.Tensor) and damping.ndim > 1:
            raise ValueError("Damping must be a float or a 1-D tensor.")
        g = self.linearization.g
        y = None
        if not getattr(self, "_unfused_forward", False):
            try:
                y = self.factorize(damping, ellipsoidal_damping, damping_eps, rhs=g)
            except RuntimeError as e:
                # the fused forward substitution keeps y_0:j in LDS next to the diagonal tile: n <= ~28 k in fp32, ~9 k in fp64
                # (fused diagonal kernel; ~18 k with the split one).  Beyond that: factorise without it and run both triangular
                # solves afterwards (the list-driven solves of the tile-sparse solver have no size limit).  The C side refuses
                # before launching anything, so nothing has to be undone.
                if "fused forward substitution" not in str(e):
                    raise
                self._unfused_forward = True
        delta = torch.empty_like(g)
        if y is not None:
            self._substitute(y, delta, backward_only=True)
        else:
            self.factorize(damping, ellipsoidal_damping, damping_eps, rhs=None)
            self._substitute(g, delta, backward_only=False)
        if getattr(self, "_check_singular", False):
            # dense_solver.py:91-103: items whose UNDAMPED AtA hits an exactly zero LU pivot are dropped (zero step, no failure, a
            # RuntimeWarning).  Here: a zero on diag(AtA) -- an all-zero column of A, the case A^T A produces (singular_mask).  Any
            # OTHER breakdown of the factorisation keeps its info code: the item is a FAILED solve (RuntimeError / FAIL status), as
            # it is in the reference, whose LU passes a rank-deficient matrix with a rounding residue on to torch.linalg.cholesky.
            # (Corner left different: exactly dependent columns with an exactly zero LU pivot but a non-zero diagonal -- the
            # reference zeroes the step, this solver reports the failed factorisation.)
            # Device-side select, no host sync; the warning is raised where the host looks at the solve anyway (check_info=True)
            # or, for the sync-free loop, by post_singular_warning() after the loop's own synchronisation.
            singular = self.singular_mask()
            self._last_singular = singular
            delta.masked_fill_(singular.unsqueeze(1), 0.0)
            self.info.masked_fill_(singular, 0)   # a dropped item is not a failed solve
            seen = singular.any()
            self._singular_seen = seen if getattr(self, "_singular_seen", None) is None else (self._singular_seen | seen)
            if check_info:
                self.post_singular_warning()
        if check_info:
            self.check_info()
        return delta


class HipCholeskySolver(HipCholeskyCore, LinearSolver):
    def __init__(self, objective: Objective, linearization_cls: Optional[Type[Linearization]] = None,
                 linearization_kwargs: Optional[Dict[str, Any]] = None, check_singular: bool = False, **kwargs):
        linearization_cls = linearization_cls or HipLinearization
        if not (isinstance(linearization_cls, type) and issubclass(linearization_cls, HipLinearization)):
            raise RuntimeError("HipCholeskySolver only works with theseus_amd.HipLinearization, "
                               f"but {linearization_cls} was provided.")
        LinearSolver.__init__(self, objective, linearization_cls, linearization_kwargs)
        self._check_singular = check_singular
        self._core_init()

    # theseus/optimizer/linear/dense_solver.py:84-123
    def solve(self, damping: Optional[Union[float, torch.Tensor]] = None, ellipsoidal_damping: bool = True,
              damping_eps: float = 1e-8, check_info: bool = True, **kwargs) -> torch.Tensor:
        return self._solve(damping, ellipsoidal_damping, damping_eps, check_info)
