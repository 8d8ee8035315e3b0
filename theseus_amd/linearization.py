"""`Linearization` plugin surface + the HIP implementation.

ABC contract: theseus/optimizer/linearization.py:16-87.  ``HipLinearization`` replaces
``DenseLinearization`` (theseus/optimizer/dense_linearization.py:15-77): same attributes
(``ordering, var_dims, var_start_cols, num_cols, num_rows``), same ``linearize / AtA / Atb / Av /
diagonal_scaling / hessian_approx`` -- but the dense ``A`` is never built: the fused HIP kernel
writes the block-sparse lower triangle of H = A^T A straight into a persistent dense (B,ld,ld)
buffer (zero-filled once; the pattern is fixed) and g = A^T b into (B,n).
"""
import abc
import os
from typing import List, Optional

import torch

from .core import Objective, Variable
from .packed import packed_for


class VariableOrdering:
    """Default = insertion order of objective.optim_vars (theseus/optimizer/variable_ordering.py:14-60)."""

    def __init__(self, objective: Objective, default_order: bool = True):
        self.objective = objective
        self._var_order: List[Variable] = []
        self._var_name_to_index = {}
        if default_order:
            for v in objective.optim_vars.values():
                self.append(v)

    def append(self, var: Variable):
        if var.name in self._var_name_to_index:
            raise ValueError("Tried to add a variable that was already added.")
        if var.name not in self.objective.optim_vars:
            raise ValueError("Variable is not an optimization variable for the objective.")
        self._var_order.append(var)
        self._var_name_to_index[var.name] = len(self._var_order) - 1

    def index_of(self, key: str) -> int:
        return self._var_name_to_index[key]

    def __getitem__(self, i) -> Variable:
        return self._var_order[i]

    def __iter__(self):
        return iter(self._var_order)

    def __len__(self):
        return len(self._var_order)

    def complete(self):
        return len(self._var_order) == self.objective.size_variables()


class Linearization(abc.ABC):
    # theseus/optimizer/linearization.py:18-41
    def __init__(self, objective: Objective, ordering: Optional[VariableOrdering] = None, **kwargs):
        self.objective = objective
        if ordering is None:
            ordering = VariableOrdering(objective, default_order=True)
        self.ordering = ordering
        if not self.ordering.complete():
            raise ValueError("Given variable ordering is not complete.")
        self.var_dims: List[int] = []
        self.var_start_cols: List[int] = []
        col = 0
        for var in ordering:
            self.var_start_cols.append(col)
            self.var_dims.append(var.dof())
            col += var.dof()
        self.num_cols = col
        self.num_rows = self.objective.dim()

    @abc.abstractmethod
    def _linearize_jacobian_impl(self):
        pass

    @abc.abstractmethod
    def _linearize_hessian_impl(self, _detach_hessian: bool = False):
        pass

    def linearize(self, _detach_hessian: bool = False):
        if not self.ordering.complete():
            raise RuntimeError("Attempted to linearize an objective with an incomplete variable order.")
        self._linearize_hessian_impl(_detach_hessian=_detach_hessian)

    def hessian_approx(self):
        return self.AtA

    @property
    @abc.abstractmethod
    def AtA(self) -> torch.Tensor:
        pass

    @property
    @abc.abstractmethod
    def Atb(self) -> torch.Tensor:
        pass

    @abc.abstractmethod
    def Av(self, v: torch.Tensor) -> torch.Tensor:
        pass

    @abc.abstractmethod
    def diagonal_scaling(self, v: torch.Tensor) -> torch.Tensor:
        pass


class HipLinearizationCore:
    """Back-end half of the linearization, independent of which ``Linearization`` ABC it is mixed into:
    theseus_amd's mirror (below) or the real ``theseus.optimizer.Linearization`` (theseus_amd/plugin.py)."""

    # (defaults for subclasses that take a generic path without _core_init: theseus_amd/plugin.py)
    _compact = False
    _H: Optional[torch.Tensor] = None
    Hc: Optional[torch.Tensor] = None
    hblocks = None

    def _core_init(self, objective, kernels=None, block_hessian: Optional[bool] = None):
        # the column order is the linearization's VariableOrdering (default: insertion order; theseus_amd/sparse.py passes a
        # fill-reducing one) -- the packed pose buffer is laid out in that order, so delta / retract / Jacobian columns agree
        self.packed = packed_for(objective, kernels, [v.name for v in self.ordering])
        self.K = self.packed.K
        # A^T A (undamped, lower triangle).  SE3 pose graphs on the HIP kernels keep it BLOCK-COMPACT: ``Hc`` (B, bstride), the
        # list of its non-zero 6 x 6 blocks (184 KB per problem at 256 poses / 1024 edges; the dense frame: 9.4 MB) -- assembly
        # writes whole 144-byte blocks, the Cholesky tiles gather their pieces.  ``H`` (B, ld, ld) is then materialised only if
        # somebody reads it (``AtA``).  Everything else (SE2 / SO3, the test stand-in kernels, THX_DENSE_HESSIAN=1): dense frame.
        self._H: Optional[torch.Tensor] = None
        self.Hc: Optional[torch.Tensor] = None
        self.hblocks = None                      # compiler.DeviceHessianBlocks when compact
        self.g: Optional[torch.Tensor] = None   # (B, n): A^T b
        want = block_hessian if block_hessian is not None else os.environ.get("THX_DENSE_HESSIAN", "0") != "1"
        self._compact = bool(want) and getattr(self.packed, "supports_block_hessian", lambda: False)()
        self._AtA_cache = None
        self._A = self._b = None
        self._Jblocks = None   # weighted Jacobian blocks of the current linearization (Av), on demand

    # structure exactly as the reference lays it out
    @property
    def n(self):
        return self.packed.n

    @property
    def ld(self):
        return self.packed.ld

    @property
    def H(self) -> Optional[torch.Tensor]:
        """(B, ld, ld) dense frame of the lower triangle.  With block-compact storage it is expanded on first use after a
        ``linearize()`` (thx_hblocks_expand) -- the optimiser never asks for it."""
        if self._compact and self.Hc is not None and self._H is None:
            self._H = torch.zeros(self.Hc.shape[0], self.ld, self.ld, dtype=self.Hc.dtype, device=self.Hc.device)
            self.K.hblocks_expand(self.hblocks, self.Hc, self._H)
        return self._H

    @H.setter
    def H(self, value):
        self._H = value

    @property
    def linearized(self) -> bool:
        return (self.Hc if self._compact else self._H) is not None

    def _ensure_buffers(self):
        self.packed.sync()
        B = self.packed.batch
        dev, dt = self.packed.device, self.objective.dtype
        if self._compact:
            if self.Hc is None or self.Hc.shape[0] != B or self.Hc.device != dev or self.Hc.dtype != dt:
                hb = self.packed.structure.hessian_blocks()
                self.hblocks = hb.on(dev)
                self.Hc = torch.empty(B, hb.bstride, dtype=dt, device=dev)   # every block is written in full by the assembly
                self.g = torch.empty(B, self.n, dtype=dt, device=dev)
            return
        if self._H is None or self._H.shape[0] != B or self._H.device != dev or self._H.dtype != dt:
            self._H = torch.zeros(B, self.ld, self.ld, dtype=dt, device=dev)  # zero once: fixed pattern
            self.g = torch.empty(B, self.n, dtype=dt, device=dev)

    def _materialize_A_b(self):
        """Dense A (B,m,n) and b (B,m) -- for tests / foreign consumers only
        (dense_linearization.py:29-56); the optimiser never calls this."""
        p = self.packed
        if hasattr(p, "dense_A_b"):     # generic path (theseus_amd/euclidean.py)
            self._A, self._b = p.dense_A_b()
            return
        J0, J1, eb, Jp, ep = p.jacobian_blocks()
        B, s, d = p.batch, p.structure, p.dof
        A = torch.zeros(B, p.m, p.n, dtype=J0.dtype, device=J0.device)
        b = torch.zeros(B, p.m, dtype=J0.dtype, device=J0.device)
        for e in range(s.num_edges):
            r, i, j = int(s.edge_row_start[e]), int(s.edge_i[e]), int(s.edge_j[e])
            A[:, r:r + d, d * i:d * i + d] = J0[e]
            A[:, r:r + d, d * j:d * j + d] = J1[e]
            b[:, r:r + d] = -eb[e]
        for k in range(s.num_priors):
            r, i = int(s.prior_row_start[k]), int(s.prior_pose[k])
            A[:, r:r + d, d * i:d * i + d] = Jp[k]
            b[:, r:r + d] = -ep[k]
        self._A, self._b = A, b

    def _assemble(self):
        self._ensure_buffers()
        if self._compact:
            self.packed.assemble_blocks(self.Hc, self.g)
            self._H = None      # (a dense expansion of the previous linearization is stale)
        else:
            self.packed.assemble(self._H, self.g)
        self._after_assemble()

    def _after_assemble(self):
        self._AtA_cache = None
        self._A = self._b = None
        self._Jblocks = None

    @property
    def A(self):
        if self._A is None:
            self._materialize_A_b()
        return self._A

    @property
    def b(self):
        if self._b is None:
            self._materialize_A_b()
        return self._b

    def _full_AtA(self) -> torch.Tensor:
        """Full symmetric (B,n,n); materialised only when read (LM needs just Atb / diagonal_scaling)."""
        if self._AtA_cache is None:
            Hl = torch.tril(self.H[:, :self.n, :self.n])
            self._AtA_cache = Hl + torch.tril(Hl, -1).transpose(1, 2)
        return self._AtA_cache

    def Av(self, v: torch.Tensor) -> torch.Tensor:
        """(B, n) -> (B, m), dense_linearization.py:73-74 -- from the per-cost Jacobian blocks of this linearization
        (thx_pg_jacobians, evaluated once per ``linearize()`` on first use: the variables still hold the values the
        linearization was taken at, as they do wherever the reference's optimizers call ``Av``: dogleg.py:66,
        trust_region.py:97), never through the dense ``A``."""
        if self._Jblocks is None:
            self._Jblocks = self.packed.jacobian_blocks()
        return self.packed.jacobian_times(self._Jblocks, v)

    def diagonal(self) -> torch.Tensor:
        d = torch.empty(self.g.shape[0], self.n, dtype=self.g.dtype, device=self.g.device)
        if self._compact:
            self.K.hblocks_diag(self.hblocks, self.Hc, d)
        else:
            self.K.diag(self._H, self.n, d)
        return d

    def diagonal_scaling(self, v: torch.Tensor) -> torch.Tensor:
        return self.diagonal() * v

    def lm_accept(self, delta, damping, prev_err, new_err, ellipsoidal, accept, down, up, reject):
        """levenberg_marquardt.py:173-201 on the packed buffers (thx_lm_accept)."""
        if self._compact:
            if ellipsoidal:
                self.K.lm_accept_diag(delta, self.g, self.diagonal(), self.n, damping, prev_err, new_err, ellipsoidal, accept, down,
                                      up, reject)
            else:
                self.K.lm_accept(delta, self.g, None, self.n, damping, prev_err, new_err, False, accept, down, up, reject)
            return
        self.K.lm_accept(delta, self.g, self._H, self.n, damping, prev_err, new_err, ellipsoidal, accept, down, up, reject)


class HipLinearization(HipLinearizationCore, Linearization):
    def __init__(self, objective: Objective, ordering: Optional[VariableOrdering] = None, kernels=None,
                 block_hessian: Optional[bool] = None, **kwargs):
        """``block_hessian``: keep A^T A as the list of its non-zero blocks (default for SE3 pose graphs on the HIP kernels;
        False: the dense (B, ld, ld) frame)."""
        Linearization.__init__(self, objective, ordering)
        self._core_init(objective, kernels, block_hessian)

    def _linearize_jacobian_impl(self):
        self._materialize_A_b()

    def _linearize_hessian_impl(self, _detach_hessian: bool = False):
        self._assemble()

    @property
    def AtA(self) -> torch.Tensor:
        return self._full_AtA()

    @property
    def Atb(self) -> torch.Tensor:
        return self.g.unsqueeze(2)
