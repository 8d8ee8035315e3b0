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
