"""Tile-sparse Cholesky solver for LARGE pose graphs: the functional analogue of the reference's ``BaspachoSparseSolver``
(theseus/optimizer/linear/baspacho_sparse_solver.py:23-148 -- symbolic analysis of the block pattern at construction,
numeric factorisation + solve per iteration, damping added to the diagonal, the factor kept for the backward pass).

Dense ``n^3/3`` stops scaling beyond ~2-4 k poses (evaluations/pose_graph_synthetic.sh sweeps to 4096).  Here the
"supernodes" are the 128-wide tile columns of the MFMA Cholesky (csrc/chol_kernels.hip):

* a fill-reducing VARIABLE ORDERING (reverse Cuthill-McKee on the pose graph: a SLAM graph -- odometry chain + local loop
  closures -- becomes banded) is handed to the linearization as its ``VariableOrdering``: the packed pose buffer, the
  Hessian columns and ``delta`` all follow it (the reference's solver permutes internally, baspacho_sparse_solver.py:72-93);
* the SYMBOLIC factorisation runs once on the host at tile granularity (``tile_pattern``): which tiles of ``L`` fill in,
  and for every such tile the block columns its K-loop has to visit;
* the NUMERIC factorisation is ``thx_chol_factor_sparse``: the same kernels as the dense solver, launched over the
  non-zero tiles only, each K-loop walking its list -- the work follows the fill instead of ``n^3/3``.  Storage stays the
  dense row-major frame (288 GB of HBM hold batch 64 of n = 12288 in fp32); tiles outside the pattern are never touched.
"""
from typing import Any, Dict, List, Optional, Sequence, Tuple, Type, Union

import numpy as np
import torch

from . import _lib
from .core import Objective
from .linear_solver import HipCholeskyCore, LinearSolver
from .linearization import HipLinearization, Linearization, VariableOrdering

TILE = _lib.THX_TILE


def rcm_order(num_nodes: int, edges: Sequence[Tuple[int, int]]) -> List[int]:
    """Reverse Cuthill-McKee permutation of an undirected graph (new position k holds old node perm[k])."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    if not len(edges):
        return list(range(num_nodes))
    e = np.asarray(edges, dtype=np.int64)
    a = coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(num_nodes, num_nodes))
    return reverse_cuthill_mckee((a + a.T).tocsr(), symmetric_mode=True).tolist()


def fill_reducing_ordering(objective, ordering_cls=VariableOrdering):
    """VariableOrdering of a pose-graph objective by reverse Cuthill-McKee on (pose, pose) adjacency of its costs
    (``ordering_cls``: theseus_amd's mirror class, or the reference's th.optimizer.VariableOrdering for the plugin)."""
    names = list(objective.optim_vars.keys())
    index = {n: k for k, n in enumerate(names)}
    edges = []
    for c in objective.cost_functions.values():
        ov = c.optim_vars
        vs = [index[v.name] for v in (ov() if callable(ov) else ov)]
        edges += [(a, b) for i, a in enumerate(vs) for b in vs[i + 1:]]
    ordering = ordering_cls(objective, default_order=False)
    for k in rcm_order(len(names), edges):
        ordering.append(objective.optim_vars[names[k]])
    return ordering


class TilePattern:
    """Tile-level symbolic Cholesky of a block-sparse SPD matrix (host tables; ``c_struct(device)`` uploads them once)."""

    def __init__(self, n: int, blocks: np.ndarray, dof: int):
        """``blocks``: (nb, 2) (row, col) indices, row >= col, of the non-zero dof x dof blocks of tril(H); n = dof * #vars."""
        nt = (n + TILE - 1) // TILE
        lp = np.zeros((nt, nt), dtype=bool)
        r0, c0 = blocks[:, 0] * dof, blocks[:, 1] * dof
        for dr in (0, dof - 1):          # a block may straddle a tile boundary (128 is not a multiple of 6)
            for dc in (0, dof - 1):
                lp[(r0 + dr) // TILE, (c0 + dc) // TILE] = True
        lp |= np.eye(nt, dtype=bool)
        lp = np.tril(lp)
        self.h_tiles = int(lp.sum())
        for j in range(nt):              # symbolic elimination: column j's rows become a clique below it
            rows = np.nonzero(lp[j + 1:, j])[0] + j + 1
            if rows.size:
                lp[np.ix_(rows, rows)] |= np.tril(np.ones((rows.size, rows.size), dtype=bool))
        self.n, self.ntiles, self.lower = n, nt, lp
        col_ptr, col_row, tile_kptr, tile_k, diag_kptr, diag_k = [0], [], [0], [], [0], []
        tile_ij = []     # (i, j) of every element of tile_k, for the slot tables of the tile-packed factor
        for j in range(nt):
            dk = np.nonzero(lp[j, :j])[0]
            diag_k += dk.tolist()
            diag_kptr.append(len(diag_k))
            rows = (np.nonzero(lp[j + 1:, j])[0] + j + 1).tolist()
            klists = {i: np.nonzero(lp[i, :j] & lp[j, :j])[0].tolist() for i in rows}
            # longest K-loop first: the workgroups of a launch are dispatched in entry order, so the long ones start early and
            # the short ones fill the tail (LPT scheduling)
            # -- after tile (j + 1, j), which goes FIRST when it exists: the factorisation's look-ahead schedule launches it on
            # its own so that the diagonal phase of column j + 1 can start under the rest of column j (col_head_host)
            for i in sorted(rows, key=lambda r: (r != j + 1, -len(klists[r]), r)):
                col_row.append(i)
                tile_k += klists[i]
                tile_ij += [(i, j)] * len(klists[i])
                tile_kptr.append(len(tile_k))
            col_ptr.append(len(col_row))
        row_ptr, row_tile = [0], []     # the same pattern by rows (list-driven triangular solves)
        for i in range(nt):
            row_tile += np.nonzero(lp[i, :i])[0].tolist()
            row_ptr.append(len(row_tile))
        i32 = lambda a: np.asarray(a if len(a) else [0], dtype=np.int32)  # noqa: E731
        # tile-packed factor: slot j = diagonal tile j, slot nt + e = off-diagonal entry e = tile (col_row[e], column of e)
        slot = {(j, j): j for j in range(nt)}
        for j in range(nt):
            for e in range(col_ptr[j], col_ptr[j + 1]):
                slot[(col_row[e], j)] = nt + e
        self.nslots = nt + len(col_row)
        self.slot = slot
        tile_sa = [slot[(j, k)] for (i, j), k in zip(tile_ij, tile_k)]
        tile_sb = [slot[(i, k)] for (i, j), k in zip(tile_ij, tile_k)]
        diag_s = [slot[(j, k)] for j in range(nt) for k in diag_k[diag_kptr[j]:diag_kptr[j + 1]]]
        row_slot = [slot[(i, k)] for i in range(nt) for k in row_tile[row_ptr[i]:row_ptr[i + 1]]]
        self.tables = dict(col_ptr=i32(col_ptr), col_row=i32(col_row), tile_kptr=i32(tile_kptr), tile_k=i32(tile_k),
                           diag_kptr=i32(diag_kptr), diag_k=i32(diag_k), row_ptr=i32(row_ptr), row_tile=i32(row_tile),
                           tile_sa=i32(tile_sa), tile_sb=i32(tile_sb), diag_s=i32(diag_s), row_slot=i32(row_slot))
        self.col_count = np.ascontiguousarray(np.diff(np.asarray(col_ptr, dtype=np.int64)).astype(np.int32))
        self.col_head = np.ascontiguousarray(np.array([1 if (col_ptr[j + 1] > col_ptr[j] and col_row[col_ptr[j]] == j + 1) else 0
                                                       for j in range(nt)], dtype=np.int32))
        self.l_tiles = int(lp.sum())
        # tile products the numeric factorisation executes (K-loop tiles + one TRSM / POTRF per tile) vs the dense count
        self.tile_products = len(tile_k) + len(diag_k) + self.l_tiles
        self.dense_tile_products = sum(j * (nt - j) + (nt - j) for j in range(nt))
        # flops of the numeric factorisation per problem (tile granularity: 2 t^3 per K-loop tile product of an off-diagonal tile,
        # t^3 per one of a diagonal tile (lower half), t^3 per substitution, t^3 / 3 per diagonal factorisation; t = 128) and
        # of the dense factorisation of the same frame -- the latter is n^3 / 3 up to the padding of the last tile
        t3 = float(TILE) ** 3
        self.flops = 2 * t3 * len(tile_k) + t3 * len(diag_k) + t3 * len(col_row) + t3 / 3 * nt
        self.dense_flops = sum((nt - 1 - j) * (2 * t3 * j + t3) + t3 * j + t3 / 3 for j in range(nt))
        self._dev: Dict[str, Any] = {}

    def c_struct(self, device) -> _lib.TilePattern:
        key = str(device)
        if key not in self._dev:
            t = {k: torch.from_numpy(v).to(device) for k, v in self.tables.items()}
            c = _lib.TilePattern()
            c.ntiles = self.ntiles
            c.nslots = self.nslots
            for k, v in t.items():
                setattr(c, k, v.data_ptr())
            c.col_count_host = self.col_count.ctypes.data
            c.col_head_host = self.col_head.ctypes.data
            self._dev[key] = (c, t)
        return self._dev[key][0]


def tile_pattern(structure, dof: int) -> TilePattern:
    """Symbolic tile factorisation of a pose-graph structure (theseus_amd.compiler.PoseGraphStructure)."""
    return TilePattern(structure.num_cols, structure.lower_block_pattern(), dof)


class HipSparseCholeskyCore(HipCholeskyCore):
    """HipCholeskyCore whose factorisation follows the tile pattern of its linearization."""

    def _sparse_init(self, packed_factor: Optional[bool] = None):
        self._core_init()
        lin = self.linearization
        self.pattern = tile_pattern(lin.packed.structure, lin.packed.dof)
        # TILE-PACKED factor: L holds only the tiles of the pattern, (B, nslots, 128, 128), instead of a dense (B, ld, ld) frame
        # (n = 12288, chain graph: 12 MB instead of 604 MB per problem in fp32 -- the batch sizes of the reference's sweep fit).
        # Goes with the block-compact Hessian (thx_chol_factor_hblocks: neither H nor L is a dense frame then).
        compact = bool(getattr(lin, "_compact", False))
        if packed_factor and not compact:
            raise RuntimeError("packed_factor=True needs the block-compact Hessian (SE3 pose graphs on the HIP kernels)")
        self.packed_factor = compact if packed_factor is None else bool(packed_factor)

    def _ensure_buffers(self):
        if not self.packed_factor:
            return HipCholeskyCore._ensure_buffers(self)
        lin = self.linearization
        g = lin.g
        shape = (g.shape[0], self.pattern.nslots, TILE, TILE)
        if self.L is None or tuple(self.L.shape) != shape or self.L.device != g.device or self.L.dtype != g.dtype:
            B = g.shape[0]
            self.L = torch.zeros(shape, dtype=g.dtype, device=g.device)   # (rows of a tile beyond the matrix stay zero: never written)
            self.panels = torch.empty(B, self.pattern.ntiles, TILE, TILE, dtype=g.dtype, device=g.device)
            self._y = torch.empty(B, lin.n, dtype=g.dtype, device=g.device)
            self.info = torch.zeros(B, dtype=torch.int32, device=g.device)
            self._lam = torch.empty(B, dtype=g.dtype, device=g.device)

    def dense_factor(self) -> torch.Tensor:
        """The factor as a dense (B, ld, ld) lower-triangular frame (tests / inspection; the solver never builds it)."""
        if not self.packed_factor:
            return self.L
        lin = self.linearization
        ld = lin.ld
        out = torch.zeros(self.L.shape[0], ld, ld, dtype=self.L.dtype, device=self.L.device)
        for (i, j), slot in self.pattern.slot.items():
            r1, c1 = min(TILE * (i + 1), ld), min(TILE * (j + 1), ld)
            out[:, TILE * i:r1, TILE * j:c1] = self.L[:, slot, :r1 - TILE * i, :c1 - TILE * j]
        return out

    def factorize(self, damping: Optional[Union[float, torch.Tensor]] = None, ellipsoidal_damping: bool = True,
                  damping_eps: float = 1e-8, rhs: Optional[torch.Tensor] = None):
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
        self._factor_call(lam, ellipsoidal_damping, damping_eps, rhs, y, pattern=self.pattern)
        return y

    def solve_with_snapshot(self, snapshot, rhs: torch.Tensor) -> torch.Tensor:
        """(L L^T)^-1 rhs with a kept copy of a factor: the list-driven solves along the pattern (the copy has the solver's own
        layout -- tile-packed or dense frame -- which thx_chol_solve would misread)."""
        L, panels = snapshot
        rhs = rhs.contiguous()
        x = torch.empty_like(rhs)
        self.K.chol_solve_sparse(L, self.linearization.n, panels, rhs, x, self.pattern, backward_only=False)
        return x

    def _substitute(self, rhs, x, backward_only: bool):
        """The triangular solves over the non-zero tiles of L only (thx_chol_solve_sparse; no limit on n)."""
        self.K.chol_solve_sparse(self.L, self.linearization.n, self.panels, rhs, x, self.pattern, backward_only=backward_only)


class HipSparseCholeskySolver(HipSparseCholeskyCore, LinearSolver):
    """``linear_solver_cls`` for large pose graphs (SE3 / SE2 / SO3): tile-sparse factorisation under a fill-reducing
    variable ordering (pass ``linearization_kwargs=dict(ordering=...)`` to impose another one)."""

    def __init__(self, objective: Objective, linearization_cls: Optional[Type[Linearization]] = None,
                 linearization_kwargs: Optional[Dict[str, Any]] = None, check_singular: bool = False,
                 packed_factor: Optional[bool] = None, **kwargs):
        linearization_cls = linearization_cls or HipLinearization
        if not (isinstance(linearization_cls, type) and issubclass(linearization_cls, HipLinearization)):
            raise RuntimeError(f"HipSparseCholeskySolver only works with theseus_amd.HipLinearization, but {linearization_cls} "
                               "was provided.")
        linearization_kwargs = dict(linearization_kwargs or {})
        if linearization_kwargs.get("ordering") is None:
            linearization_kwargs["ordering"] = fill_reducing_ordering(objective)
        LinearSolver.__init__(self, objective, linearization_cls, linearization_kwargs)
        self._check_singular = check_singular
        self._sparse_init(packed_factor)

    def solve(self, damping: Optional[Union[float, torch.Tensor]] = None, ellipsoidal_damping: bool = True,
              damping_eps: float = 1e-8, check_info: bool = True, **kwargs) -> torch.Tensor:
        return self._solve(damping, ellipsoidal_damping, damping_eps, check_info)
