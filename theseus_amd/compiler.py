"""Problem compiler: objective structure -> int32 index tables for the HIP kernels.

Runs once per objective (host side, numpy).  Structure parity with the reference is bit exact:
column layout = variable order * dof (``Linearization.var_start_cols``,
theseus/optimizer/linearization.py:31-41, default ``VariableOrdering`` = insertion order,
theseus/optimizer/variable_ordering.py:19-27) and row layout = cost-function add order
(theseus/optimizer/dense_linearization.py:41-56).
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

DOF = 6


@dataclass
class PoseGraphStructure:
    """Immutable topology of a pose graph (SE3: dof 6, SE2: dof 3): P poses, E Between edges, K Difference priors."""

    num_poses: int
    edge_i: np.ndarray  # (E,) int32  v0 pose of edge e
    edge_j: np.ndarray  # (E,) int32  v1 pose of edge e
    prior_pose: np.ndarray  # (K,) int32
    # row start (in A / b / error vector) of every cost, in add order
    edge_row_start: np.ndarray = None  # (E,) int64
    prior_row_start: np.ndarray = None  # (K,) int64
    # derived CSR tables
    inc_ptr: np.ndarray = field(default=None, repr=False)
    inc_edge: np.ndarray = field(default=None, repr=False)
    inc_side: np.ndarray = field(default=None, repr=False)
    inc_other: np.ndarray = field(default=None, repr=False)
    pri_ptr: np.ndarray = field(default=None, repr=False)
    pri_id: np.ndarray = field(default=None, repr=False)
    _dev: dict = field(default_factory=dict, repr=False)
    dof: int = DOF

    @staticmethod
    def build(num_poses: int, edges: Sequence[Tuple[int, int]], priors: Sequence[int],
              edge_row_start: Optional[Sequence[int]] = None,
              prior_row_start: Optional[Sequence[int]] = None, dof: int = DOF) -> "PoseGraphStructure":
        P = int(num_poses)
        e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
        E = e.shape[0]
        pr = np.asarray(priors, dtype=np.int64).reshape(-1)
        K = pr.shape[0]
        if E and (e.min() < 0 or e.max() >= P):
            raise ValueError("edge endpoint out of range")
        if K and (pr.min() < 0 or pr.max() >= P):
            raise ValueError("prior pose out of range")
        if E and np.any(e[:, 0] == e[:, 1]):
            raise ValueError("Between cost with v0 is v1 (self loop) is not supported")
        if edge_row_start is None:
            edge_row_start = dof * np.arange(E)
        if prior_row_start is None:
            prior_row_start = dof * (E + np.arange(K))
        # incident-edge CSR per pose, entries sorted by (other endpoint, edge id)
        ent_pose = np.concatenate([e[:, 0], e[:, 1]])
        ent_other = np.concatenate([e[:, 1], e[:, 0]])
        ent_edge = np.concatenate([np.arange(E), np.arange(E)])
        ent_side = np.concatenate([np.zeros(E, np.int64), np.ones(E, np.int64)])
        order = np.lexsort((ent_edge, ent_other, ent_pose))
        inc_ptr = np.zeros(P + 1, np.int64)
        np.add.at(inc_ptr, ent_pose + 1, 1)
        inc_ptr = np.cumsum(inc_ptr)
        porder = np.argsort(pr, kind="stable")
        pri_ptr = np.zeros(P + 1, np.int64)
        np.add.at(pri_ptr, pr + 1, 1)
        pri_ptr = np.cumsum(pri_ptr)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
        return PoseGraphStructure(
            num_poses=P, edge_i=i32(e[:, 0]), edge_j=i32(e[:, 1]), prior_pose=i32(pr),
            edge_row_start=np.asarray(edge_row_start, np.int64), prior_row_start=np.asarray(prior_row_start, np.int64),
            inc_ptr=i32(inc_ptr), inc_edge=i32(ent_edge[order]), inc_side=i32(ent_side[order]),
            inc_other=i32(ent_other[order]), pri_ptr=i32(pri_ptr), pri_id=i32(porder), dof=int(dof),
        )

    # ---- sizes -------------------------------------------------------------------------------
    @property
    def num_edges(self) -> int:
        return int(self.edge_i.shape[0])

    @property
    def num_priors(self) -> int:
        return int(self.prior_pose.shape[0])

    @property
    def num_cols(self) -> int:
        return self.dof * self.num_poses

    @property
    def num_rows(self) -> int:
        return self.dof * (self.num_edges + self.num_priors)

    @property
    def var_start_cols(self) -> List[int]:
        return [self.dof * k for k in range(self.num_poses)]

    def lower_block_pattern(self) -> np.ndarray:
        """(nblocks, 2) sorted unique (row pose, col pose) of the non-zero 6x6 blocks of tril(H)."""
        blocks = {(p, p) for p in range(self.num_poses)}
        for i, j in zip(self.edge_i.tolist(), self.edge_j.tolist()):
            blocks.add((max(i, j), min(i, j)))
        return np.array(sorted(blocks), dtype=np.int64)

    def hessian_blocks(self) -> "HessianBlocks":
        """Block-compact layout of tril(H) for this structure (built once)."""
        hb = self._dev.get("__hblocks__")
        if hb is None:
            hb = self._dev["__hblocks__"] = HessianBlocks(self)
        return hb

    # ---- device side ---------------------------------------------------------------------------
    def on(self, device) -> "DeviceStructure":
        key = str(device)
        if key not in self._dev:
            self._dev[key] = DeviceStructure(self, device)
        return self._dev[key]


class DeviceStructure:
    """The int32 tables on one device + the ctypes struct that points at them."""

    _FIELDS = ["edge_i", "edge_j", "inc_ptr", "inc_edge", "inc_side", "inc_other", "prior_pose", "pri_ptr", "pri_id"]

    def __init__(self, s: PoseGraphStructure, device):
        self.host = s
        self.device = torch.device(device)
        self.t = {}
        for f in self._FIELDS:
            a = getattr(s, f)
            if a.size == 0:
                a = np.zeros(1, np.int32)  # keep a valid pointer
            self.t[f] = torch.from_numpy(a.copy()).to(self.device)
        c = _lib.PGStructure()
        c.num_poses, c.num_edges, c.num_priors = s.num_poses, s.num_edges, s.num_priors
        for f in self._FIELDS:
            setattr(c, f, self.t[f].data_ptr())
        self.c = c


TILE = _lib.THX_TILE


class HessianBlocks:
    """Block-compact storage of tril(H) (include/theseus_hip.h: thx_hblock_layout): the non-zero dof x dof blocks -- one per
    variable (diagonal) and one per connected variable pair -- ordered by the 128 x 128 Cholesky tile of their top-left element,
    then by (row variable, column variable), so that the blocks of a tile are one contiguous run of the list; per lower tile the
    PIECES that fall into it (a block straddles up to four tiles: 128 is not a multiple of 6)."""

    def __init__(self, s: PoseGraphStructure):
        d, P = s.dof, s.num_poses
        pat = s.lower_block_pattern()                                   # (nb, 2) sorted by (p, q), q <= p
        p, q = pat[:, 0], pat[:, 1]
        order = np.lexsort((q, p, (d * q) // TILE, (d * p) // TILE))    # by (tile row, tile col, p, q)
        self.blocks = pat[order]
        self.nblocks, self.bd, self.nvars = int(pat.shape[0]), d, P
        self.ntiles = (d * P + TILE - 1) // TILE
        ident = {(int(a), int(b)): k for k, (a, b) in enumerate(self.blocks.tolist())}
        self.diag_blk = np.array([ident[(k, k)] for k in range(P)], dtype=np.int32)
        pose_of_entry = np.repeat(np.arange(P), np.diff(s.inc_ptr))
        self.inc_blk = np.array([ident[(int(a), int(o))] if o < a else -1 for a, o in zip(pose_of_entry.tolist(), s.inc_other.tolist())],
                                dtype=np.int32)
        # pieces per lower tile t = i (i + 1) / 2 + j
        per_tile = [[] for _ in range(self.ntiles * (self.ntiles + 1) // 2)]
        for k, (a, b) in enumerate(self.blocks.tolist()):
            r, c = d * a, d * b
            for ti in sorted({r // TILE, (r + d - 1) // TILE}):
                for tj in sorted({c // TILE, (c + d - 1) // TILE}):
                    if tj > ti:
                        continue   # (the part of a straddling DIAGONAL block above the tile diagonal: never read)
                    per_tile[ti * (ti + 1) // 2 + tj].append((k, r - TILE * ti, c - TILE * tj))
        self.tile_ptr = np.zeros(len(per_tile) + 1, dtype=np.int32)
        self.tile_ptr[1:] = np.cumsum([len(x) for x in per_tile])
        flat = [x for lst in per_tile for x in lst]
        self.piece_blk = np.array([x[0] for x in flat], dtype=np.int32)
        self.piece_rc = np.array([((x[1] & 0xFFFF) << 16) | (x[2] & 0xFFFF) for x in flat], dtype=np.int64).astype(np.uint32).view(np.int32)
        self.elems = self.nblocks * d * d
        self.bstride = (self.elems + 3) // 4 * 4     # elements per problem (16-byte aligned in fp32)
        self._dev = {}

    def on(self, device) -> "DeviceHessianBlocks":
        key = str(device)
        if key not in self._dev:
            self._dev[key] = DeviceHessianBlocks(self, device)
        return self._dev[key]

    # ---- host-side reference of the layout (tests) ---------------------------------------------------------------
    def pack_dense(self, H: np.ndarray) -> np.ndarray:
        """(B, >= n, >= n) dense (lower) -> (B, bstride) block list."""
        B, d = H.shape[0], self.bd
        out = np.zeros((B, self.bstride), dtype=H.dtype)
        for k, (a, b) in enumerate(self.blocks.tolist()):
            out[:, k * d * d:(k + 1) * d * d] = H[:, d * a:d * a + d, d * b:d * b + d].reshape(B, -1)
        return out

    def expand(self, Hc: np.ndarray, ld: int) -> np.ndarray:
        """(B, bstride) -> dense (B, ld, ld) the way the device kernels gather it: tile by tile, piece by piece."""
        B, d = Hc.shape[0], self.bd
        H = np.zeros((B, ld, ld), dtype=Hc.dtype)
        for ti in range(self.ntiles):
            for tj in range(ti + 1):
                t = ti * (ti + 1) // 2 + tj
                for pc in range(self.tile_ptr[t], self.tile_ptr[t + 1]):
                    rc = int(self.piece_rc[pc]) & 0xFFFFFFFF
                    r0, c0 = np.array(rc >> 16, dtype=np.uint16).view(np.int16), np.array(rc & 0xFFFF, dtype=np.uint16).view(np.int16)
                    blk = Hc[:, self.piece_blk[pc] * d * d:(self.piece_blk[pc] + 1) * d * d].reshape(B, d, d)
                    for e in range(d * d):
                        r, c = int(r0) + e // d, int(c0) + e % d
                        if 0 <= r < TILE and 0 <= c < TILE and TILE * ti + r < ld and TILE * tj + c < ld:
                            H[:, TILE * ti + r, TILE * tj + c] = blk[:, e // d, e % d]
        return H


class DeviceHessianBlocks:
    _FIELDS = ["diag_blk", "inc_blk", "tile_ptr", "piece_blk", "piece_rc"]

    def __init__(self, hb: HessianBlocks, device):
        self.host = hb
        self.t = {}
        for f in self._FIELDS:
            a = getattr(hb, f)
            if a.size == 0:
                a = np.zeros(1, np.int32)
            self.t[f] = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        c = _lib.HBlockLayout()
        c.nblocks, c.bd, c.nvars, c.ntiles = hb.nblocks, hb.bd, hb.nvars, hb.ntiles
        for f in self._FIELDS:
            setattr(c, f, self.t[f].data_ptr())
        c.max_tile_pieces = _lib.max_offdiag_tile_pieces(hb.tile_ptr, hb.ntiles)
        self.c = c
