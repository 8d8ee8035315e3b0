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
import os
import re
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


def _objective_graph(objective):
    names = list(objective.optim_vars.keys())
    index = {n: k for k, n in enumerate(names)}
    edges = []
    for c in objective.cost_functions.values():
        ov = c.optim_vars
        vs = [index[v.name] for v in (ov() if callable(ov) else ov)]
        edges += [(a, b) for i, a in enumerate(vs) for b in vs[i + 1:]]
    return names, edges


LEVEL_SCHEDULE_MAX_BATCH = 1024   # ordering="auto": from this batch size on the column-by-column schedule under RCM (see level_ordering)


def fill_reducing_ordering(objective, ordering_cls=VariableOrdering):
    """VariableOrdering of a pose-graph objective by reverse Cuthill-McKee on (pose, pose) adjacency of its costs
    (``ordering_cls``: theseus_amd's mirror class, or the reference's th.optimizer.VariableOrdering for the plugin)."""
    names, edges = _objective_graph(objective)
    ordering = ordering_cls(objective, default_order=False)
    for k in rcm_order(len(names), edges):
        ordering.append(objective.optim_vars[names[k]])
    return ordering


def level_ordering(objective, ordering_cls=VariableOrdering, method: str = "auto", batch_hint: Optional[int] = None):
    """(VariableOrdering, variables per tile, info) for the LEVEL-SCHEDULED solver: tile-level nested dissection of the
    objective's variable graph (tile_nested_dissection) -- or (RCM ordering, None, info) when the level schedule does not apply:
    variables that are not all SE3 (the block-compact Hessian it reads is implemented for SE3 pose graphs), or ``method="rcm"``,
    or the time model ranks the plain band order first (then the column-by-column schedule with its look-ahead is the better
    one).  ``batch_hint``: the batch size the model ranks the candidate orders at (default: the objective's, else 64)."""
    names, edges = _objective_graph(objective)
    vs = [objective.optim_vars[n] for n in names]
    dofs = {int(v.dof()) for v in vs}
    se3 = bool(vs) and all(type(v).__name__ == "SE3" for v in vs) and dofs == {6}
    if method == "rcm" or not se3 or TILE // 6 >= len(vs):   # (one tile: nothing to dissect)
        return fill_reducing_ordering(objective, ordering_cls), None, dict(method="rcm")
    if batch_hint is None:
        batch_hint = getattr(objective, "batch_size", None) or 64
    if method == "auto" and int(batch_hint) >= LEVEL_SCHEDULE_MAX_BATCH:
        # the batch alone fills the chip several times over: the column-by-column schedule with its two half-batch streams (one
        # half's diagonal phases under the other half's off-diagonal launches) and unpadded tiles is the faster one -- headline
        # topology at batch 4096: 100.5 k problem-iterations/s against 94.5 k through the level schedule (13 padded tiles instead
        # of 12, one stream; profiles/r6/q_ab_tile_sparse_ordering_batch_4096.txt).  The time model does not see those streams.
        return fill_reducing_ordering(objective, ordering_cls), None, dict(method="rcm", reason=f"batch >= {LEVEL_SCHEDULE_MAX_BATCH}")
    order, counts, info = tile_nested_dissection(len(names), edges, TILE // 6, batch_hint=int(batch_hint), method=method)
    if info["method"] == "band" and method == "auto":
        return fill_reducing_ordering(objective, ordering_cls), None, dict(method="rcm", model=info)
    ordering = ordering_cls(objective, default_order=False)
    for k in order:
        ordering.append(objective.optim_vars[names[k]])
    return ordering, counts, info


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


# ------------------------------------------------------------------------------------------------------------------------------
# Elimination-tree parallelism: nested dissection at TILE granularity + level-scheduled launches (thx_chol_factor_levels)
# ------------------------------------------------------------------------------------------------------------------------------
def _components(adj: np.ndarray, nodes: np.ndarray) -> List[np.ndarray]:
    """Connected components of the subgraph induced by ``nodes`` (adj: dense bool matrix)."""
    left = set(int(v) for v in nodes)
    inside = np.zeros(adj.shape[0], dtype=bool)
    inside[nodes] = True
    out = []
    while left:
        seed = min(left)
        comp, frontier = [seed], [seed]
        left.discard(seed)
        seen = np.zeros(adj.shape[0], dtype=bool)
        seen[seed] = True
        while frontier:
            nb = np.nonzero(adj[frontier].any(axis=0) & inside & ~seen)[0]
            seen[nb] = True
            frontier = nb.tolist()
            comp += frontier
            left.difference_update(frontier)
        out.append(np.array(sorted(comp), dtype=np.int64))
    return out


def _bfs_levels(adj: np.ndarray, nodes: np.ndarray, root: int) -> List[np.ndarray]:
    inside = np.zeros(adj.shape[0], dtype=bool)
    inside[nodes] = True
    seen = np.zeros(adj.shape[0], dtype=bool)
    seen[root] = True
    levels, frontier = [np.array([root], dtype=np.int64)], [root]
    while True:
        nb = np.nonzero(adj[frontier].any(axis=0) & inside & ~seen)[0]
        if not nb.size:
            return levels
        seen[nb] = True
        levels.append(nb)
        frontier = nb.tolist()


def _pseudo_peripheral(adj: np.ndarray, nodes: np.ndarray) -> int:
    """A node of (nearly) maximal eccentricity in the connected subgraph ``nodes`` (George-Liu: repeat BFS from the lowest-degree
    node of the last level while the level structure gets deeper)."""
    inside = np.zeros(adj.shape[0], dtype=bool)
    inside[nodes] = True
    deg = lambda v: int((adj[v] & inside).sum())  # noqa: E731
    root = int(min(nodes.tolist(), key=deg))
    depth = -1
    for _ in range(8):
        lv = _bfs_levels(adj, nodes, root)
        if len(lv) <= depth:
            break
        depth = len(lv)
        root = int(min(lv[-1].tolist(), key=deg))
    return root


def minimum_degree(adj: np.ndarray, nodes: Optional[Sequence[int]] = None) -> List[int]:
    """Greedy minimum-degree elimination order of ``nodes`` (default: all) in the graph ``adj`` (dense symmetric bool): the node
    with the fewest neighbours goes next (ties: lowest index), its neighbours become a clique.  Nodes outside ``nodes`` stay in
    the graph as neighbours that are never eliminated -- the separators a leaf of the nested dissection hangs on: a chain with
    one such anchor is numbered from its far end towards it (no fill)."""
    n = adj.shape[0]
    nb = [set(np.nonzero(adj[v])[0].tolist()) for v in range(n)]
    todo = set(range(n) if nodes is None else (int(v) for v in nodes))
    order: List[int] = []
    while todo:
        v = min(todo, key=lambda u: (len(nb[u]), u))
        todo.discard(v)
        order.append(v)
        ring = nb[v]
        for a in ring:
            nb[a].discard(v)
            nb[a] |= ring - {a}
        nb[v] = set()
    return order


def nested_dissection(adj: np.ndarray, leaf: int = 1) -> List[int]:
    """Nested-dissection elimination order of a graph (dense symmetric bool adjacency, no self loops): recursive bisection by the
    middle level of a BFS level structure rooted at a pseudo-peripheral node (George's automatic nested dissection), the
    separator thinned to the nodes that really touch the far side.  Subgraphs of <= ``leaf`` nodes are numbered in index order
    (the caller's band order).  Children before separators: the order's elimination tree has depth ~log2 on chain-/mesh-like
    graphs -- that depth is the number of dependent launch pairs of thx_chol_factor_levels.  Leaves of several nodes are numbered
    by minimum degree with the separators above them kept in the graph (a leaf chain runs TOWARDS its separator)."""
    order: List[int] = []
    pending: List[np.ndarray] = []      # leaves, numbered once the whole separator structure is known

    def rec(nodes: np.ndarray):
        if nodes.size <= max(leaf, 1):
            if nodes.size > 1:
                pending.append(nodes)
                order.extend([-1 - len(pending)] * nodes.size)    # placeholder, filled in below
            else:
                order.extend(sorted(nodes.tolist()))
            return
        comps = _components(adj, nodes)
        if len(comps) > 1:
            for c in comps:
                rec(c)
            return
        lv = _bfs_levels(adj, nodes, _pseudo_peripheral(adj, nodes))
        if len(lv) < 3:       # (a clique-like subgraph: no separator worth having)
            order.extend(sorted(nodes.tolist()))
            return
        sizes = np.array([x.size for x in lv])
        cum = np.cumsum(sizes)
        total = int(cum[-1])
        # separator = level m (1 <= m <= len - 2): balance the two sides, prefer thin levels
        cost = [max(cum[m - 1], total - cum[m]) + 2 * sizes[m] for m in range(1, len(lv) - 1)]
        m = 1 + int(np.argmin(cost))
        far = np.concatenate(lv[m + 1:])
        far_mask = np.zeros(adj.shape[0], dtype=bool)
        far_mask[far] = True
        touches = (adj[lv[m]] & far_mask).any(axis=1)
        sep = lv[m][touches]
        near = np.concatenate(lv[:m] + [lv[m][~touches]])
        rec(np.sort(near))
        rec(np.sort(far))
        order.extend(sorted(sep.tolist()))

    rec(np.arange(adj.shape[0], dtype=np.int64))
    if pending:
        out: List[int] = []
        k = 0
        while k < len(order):
            if order[k] >= 0:
                out.append(order[k])
                k += 1
                continue
            leaf_nodes = pending[-2 - order[k]]
            # the leaf's own nodes by minimum degree inside (leaf + everything numbered AFTER it: its separators)
            later = set(v for v in order[k + leaf_nodes.size:] if v >= 0)
            for q in range(k + leaf_nodes.size, len(order)):
                if order[q] < 0:
                    later |= set(pending[-2 - order[q]].tolist())
            keep = np.zeros(adj.shape[0], dtype=bool)
            keep[list(later)] = True
            keep[leaf_nodes] = True
            sub = adj & keep[:, None] & keep[None, :]
            out.extend(minimum_degree(sub, leaf_nodes.tolist()))
            k += leaf_nodes.size
        order = out
    return order


def _symbolic_tiles(lp: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Symbolic Cholesky of a lower bool tile pattern (in place fill) -> (filled pattern, level of every column in the elimination
    tree: 0 for a column that depends on no other, else 1 + the deepest column its row panel touches)."""
    nt = lp.shape[0]
    lp = np.tril(lp | np.eye(nt, dtype=bool))
    for j in range(nt):
        rows = np.nonzero(lp[j + 1:, j])[0] + j + 1
        if rows.size:
            lp[np.ix_(rows, rows)] |= np.tril(np.ones((rows.size, rows.size), dtype=bool))
    level = np.zeros(nt, dtype=np.int64)
    for j in range(nt):
        k = np.nonzero(lp[j, :j])[0]
        if k.size:
            level[j] = level[k].max() + 1
    return lp, level


def _subtree_groups(lp: np.ndarray) -> np.ndarray:
    """Stream group of every block column for the level schedule's two chain streams: walk the tile elimination tree (parent of
    column j = its first non-zero row tile below the diagonal) down from the root(s) while a node has one child -- that path is
    the TRUNK (group -1: launched after both streams have joined); at the first node with several children their subtrees are
    dealt to groups 0 / 1, largest first onto the lighter group.  Subtrees do not see each other (a column's K-lists name its
    descendants only), so the two groups' launches need no ordering against each other.  A tree that never branches (a band): all
    columns in group 0, no trunk."""
    nt = lp.shape[0]
    parent = np.full(nt, -1, dtype=np.int64)
    for j in range(nt):
        rows = np.nonzero(lp[j + 1:, j])[0]
        if rows.size:
            parent[j] = j + 1 + rows[0]
    children: List[List[int]] = [[] for _ in range(nt)]
    roots = []
    for j in range(nt):
        (children[parent[j]] if parent[j] >= 0 else roots).append(j)
    size = np.ones(nt, dtype=np.int64)
    for j in range(nt):                      # (children have smaller indices than their parent)
        if parent[j] >= 0:
            size[parent[j]] += size[j]
    group = np.zeros(nt, dtype=np.int64)
    heads, trunk = roots, []
    while len(heads) == 1:
        trunk.append(heads[0])
        heads = children[heads[0]]
    if not heads:                            # a chain: nothing to split
        return group
    group[trunk] = -1
    load = [0, 0]
    for h in sorted(heads, key=lambda v: -int(size[v])):
        g = 0 if load[0] <= load[1] else 1
        load[g] += int(size[h])
        stack = [h]
        while stack:
            v = stack.pop()
            group[v] = g
            stack.extend(children[v])
    if min(load) == 0:                       # everything landed on one stream
        group[group >= 0] = 0
    return group


def _schedule_cost_us(lp: np.ndarray, level: np.ndarray, batch: int) -> float:
    """Time model of one level-scheduled linear solve (factorisation with the fused forward substitution + the backward solve), in
    microseconds -- only used to RANK candidate orderings.  Per level three launches, each the larger of a latency term (the
    launch is a few workgroups: one workgroup's own duration) and a throughput term (workgroups x cost per workgroup with the
    chip full).  Constants from kernel traces of tools/bench_sparse.py at 4096 poses (profiles/r6/: batch 8 -- a fused diagonal
    workgroup alone 36 us + 5.5 us per K-list tile, an off-diagonal one 15 us + 7 us per tile, a backward-solve row 19 us;
    batch 256 -- 0.0575 us per diagonal tile through SYRK + potrf, 0.033 us per off-diagonal workgroup + 0.022 us per K-list
    tile, 0.012 us per 64 KB tile the backward solve streams); checked against the measured factorisations of six orderings at
    batch 8 / 64 / 256 (profiles/r6/d_orderings.txt: the ranking agrees, the times within ~15 %)."""
    nt = lp.shape[0]
    rowcount = lp.sum(axis=1) - 1                     # off-diagonal tiles of block row j = K-list length of diagonal tile j
    total = 0.0
    for lv in range(int(level.max()) + 1):
        cols = np.nonzero(level == lv)[0]
        kd = rowcount[cols]
        total += 3.0 + max(36.0 + 5.5 * float(kd.max()), batch * float((0.0575 + 0.01 * kd).sum()))
        ko = []
        for j in cols:
            rows = np.nonzero(lp[j + 1:, j])[0] + j + 1
            if rows.size:
                ko.append((lp[rows, :j] & lp[j, :j]).sum(axis=1))
        if ko:
            ko = np.concatenate(ko)
            total += 3.0 + max(15.0 + 7.0 * float(ko.max()), batch * float((0.033 + 0.022 * ko).sum()))
        total += 3.0 + max(19.0, batch * 0.012 * float(cols.size + kd.sum()))      # backward solve: panels + the rows' tiles
    return total


def tile_nested_dissection(num_vars: int, edges: Sequence[Tuple[int, int]], vars_per_tile: int, batch_hint: int = 64,
                           method: str = "auto") -> Tuple[List[int], List[int], Dict[str, Any]]:
    """Variable order + tile partition for the level-scheduled solver.

    1. reverse Cuthill-McKee on the variable graph, cut into CLUSTERS of ``vars_per_tile`` consecutive variables (= one 128-wide
       Cholesky tile each, padded: no variable straddles a tile) -- clusters of a banded order are compact;
    2. candidate cluster orders on the QUOTIENT graph: the band order itself (a chain of dependent columns, least fill) and nested
       dissections with leaf chains of 1, 2, 4 clusters (log-depth elimination trees, more fill), ranked by a time model of the
       level schedule at ``batch_hint`` problems (``method``: "auto" | "band" | "nd");
    3. the chosen order's columns are renumbered LEVEL BY LEVEL of its elimination tree (a topological order of the same tree:
       same fill), which is what thx_chol_factor_levels launches.
    Returns (variable order, variables per tile, info)."""
    perm = rcm_order(num_vars, edges)
    vpt = int(vars_per_tile)
    nc = (num_vars + vpt - 1) // vpt
    cluster_of = np.empty(num_vars, dtype=np.int64)
    cluster_of[np.asarray(perm, dtype=np.int64)] = np.arange(num_vars) // vpt
    adj = np.zeros((nc, nc), dtype=bool)
    if len(edges):
        e = np.asarray(edges, dtype=np.int64)
        a, b = cluster_of[e[:, 0]], cluster_of[e[:, 1]]
        adj[a, b] = True
        adj[b, a] = True
    np.fill_diagonal(adj, False)
    cands: Dict[str, List[int]] = {}
    if method in ("auto", "band"):
        cands["band"] = list(range(nc))
    # leaves of 1 cluster: the deepest dissection (log-depth tree, every interior cluster pays the fill of a parallel chain
    # elimination: ~2.6 x the band's arithmetic on a chain-like graph); leaves of nc / 2, nc / 4, nc / 8 clusters: two, four, eight
    # long chains running towards a few separators -- two ("twisted factorisation") cost no fill at all, at half the band's depth
    leaves = sorted({1, 4} | {max(1, -(-nc // d)) for d in (2, 4, 8, 16)})
    m = re.fullmatch(r"nd(\d+)", method)
    for leaf in ([int(m.group(1))] if m else leaves):
        if m or method in ("auto", "nd"):
            cands[f"nd{leaf}"] = nested_dissection(adj, leaf=leaf)
    if method in ("auto", "md"):
        cands["md"] = minimum_degree(adj)
    if not cands:
        raise ValueError(f"unknown ordering method {method!r}")
    best, model = None, {}
    for name, order in cands.items():
        o = np.asarray(order, dtype=np.int64)
        lp, level = _symbolic_tiles(np.tril(adj[np.ix_(o, o)]))
        cost = _schedule_cost_us(lp, level, batch_hint)
        model[name] = dict(model_us=round(cost, 1), levels=int(level.max()) + 1, l_tiles=int(lp.sum()))
        if best is None or cost < best[0]:
            best = (cost, name, o, level, int(lp.sum()))
    cost, name, o, level, l_tiles = best
    o = o[np.argsort(level, kind="stable")]          # level by level (stable: a level keeps the candidate's internal order)
    # ... and inside a level by stream group (LevelPattern splits every level into one launch pair per group)
    lp_l, level_l = _symbolic_tiles(np.tril(adj[np.ix_(o, o)]))
    grp = _subtree_groups(lp_l)
    o = o[np.lexsort((np.arange(nc), grp, level_l))]
    members = [[] for _ in range(nc)]
    for k, v in enumerate(perm):
        members[k // vpt].append(int(v))
    order = [v for c in o.tolist() for v in members[c]]
    counts = [len(members[c]) for c in o.tolist()]
    info = dict(method=name, levels=int(level.max()) + 1, tiles=nc, l_tiles=l_tiles, model_us=cost, candidates=model)
    return order, counts, info


class LevelPattern:
    """Tile-level symbolic Cholesky for thx_chol_factor_levels / thx_chol_solve_levels (include/theseus_hip.h:
    thx_tile_pattern + thx_level_schedule): PADDED tiles (tile j holds ``tile_count[j]`` whole variables, the rest of its 128
    rows / columns is identity padding), block columns numbered level by level of the tile elimination tree, every level's
    off-diagonal entries sorted longest K-list first; the factor is tile-packed (slot j = diagonal tile j, slot ntiles + e =
    entry e)."""

    def __init__(self, blocks: np.ndarray, dof: int, tile_count: Sequence[int]):
        """``blocks``: (nb, 2) (row, col) VARIABLE positions (row >= col) of the non-zero blocks of tril(H) in the linearization's
        column order; ``tile_count``: consecutive variables per tile."""
        counts = np.asarray(tile_count, dtype=np.int64)
        if counts.min() < 1 or (counts * dof).max() > TILE:
            raise ValueError("every tile holds between 1 and TILE // dof variables")
        nt, P = counts.size, int(counts.sum())
        start = np.concatenate([[0], np.cumsum(counts)])
        tile_of = np.repeat(np.arange(nt), counts)
        slot_of = np.arange(P) - start[tile_of]
        self.dof, self.ntiles, self.nvars = dof, nt, P
        self.n, self.npad = dof * P, TILE * nt
        self.tile_count, self.tile_of, self.slot_of = counts, tile_of, slot_of
        lp0 = np.zeros((nt, nt), dtype=bool)
        lp0[tile_of[blocks[:, 0]], tile_of[blocks[:, 1]]] = True
        self.h_tiles = int(np.tril(lp0 | np.eye(nt, dtype=bool)).sum())
        lp, level = _symbolic_tiles(lp0)
        if np.any(np.diff(level) < 0):
            raise ValueError("block columns must be numbered level by level of the tile elimination tree "
                             "(theseus_amd.sparse.tile_nested_dissection produces such an order)")
        self.lower, self.tree_level = lp, level
        self.tree_levels = int(level.max()) + 1
        self.tree_level_col = np.ascontiguousarray(np.searchsorted(level, np.arange(self.tree_levels + 1)).astype(np.int32))
        # LAUNCH levels: a tree level is split by stream group when its columns are sorted by group (tile_nested_dissection does
        # that) -- group 0 on the caller's stream, group 1 on the library's second stream, the trunk (-1) on the caller's stream
        # after the join; the groups' subtrees do not see each other, so a level's two launch pairs are independent and the two
        # streams drift apart: one chain's diagonal phase (one busy wave per workgroup) runs beside the other's off-diagonal tiles
        group = _subtree_groups(lp)
        key = level * 3 + (group + 1)
        # ... only for CHAIN-like subtrees (at most two columns of a group on a tree level: the two-chain orders of a band, e.g. the
        # reduced camera system of bundle adjustment): a bushy level cut in two is two half-size launches, which costs more than
        # the overlap returns (4096 poses under nd1 at batch 8: factor 0.84 -> 0.98 ms with the split, 1.39 without the streams).
        # Same-box A/B against ONE launch pair per tree level (profiles/r6/u_ab_subtree_streams.txt, three interleaved rounds):
        # bundle adjustment under nd13: factor 6.66 - 6.73 vs 6.85 - 6.90 ms, but 11.08 - 11.42 vs 11.19 - 11.25 ms per linear solve
        # inside the LM loop; 4096 poses under two chains (nd98) at batch 256: factor 7.91 vs 7.43 ms.  The trunk of the tree (the
        # separator: seven serial block columns of the camera system) has nothing to overlap with.  Kept behind the switch.
        widest = max((int(((level == lv) & (group == g)).sum()) for lv in range(int(level.max()) + 1) for g in (0, 1)), default=0)
        if (np.any(np.diff(key) < 0) or widest > 2 or not (group == 1).any()     # (else: levels only, everything on one stream)
                or os.environ.get("THX_LEVEL_SUBTREES", "0") != "1"):   # MEASURED, NOT A WIN: off unless THX_LEVEL_SUBTREES=1
            group = np.zeros(nt, dtype=np.int64)
            key = level * 3 + 1
        _, plevel = np.unique(key, return_inverse=True)
        self.group = group
        level = plevel.astype(np.int64)
        self.level = level
        nlev = int(level.max()) + 1
        self.nlevels = nlev
        level_col = np.searchsorted(level, np.arange(nlev + 1)).astype(np.int32)
        # stream of every launch level: 0 | 1, +4 on the first trunk level (both streams join in front of it)
        lstream = np.array([max(int(group[level_col[lv]]), 0) for lv in range(nlev)], dtype=np.int32)
        trunk_lv = [lv for lv in range(nlev) if group[level_col[lv]] < 0]
        if trunk_lv and (lstream == 1).any():
            lstream[trunk_lv[0]] |= 4
        self.level_stream = np.ascontiguousarray(lstream)
        self.two_streams = bool((lstream & 3 == 1).any())
        diag_kptr, diag_k = [0], []
        for j in range(nt):
            diag_k += np.nonzero(lp[j, :j])[0].tolist()
            diag_kptr.append(len(diag_k))
        col_row, ent_col, tile_kptr, tile_k, tile_ij, level_ent = [], [], [0], [], [], [0]
        for lv in range(nlev):
            ents = []
            for j in range(level_col[lv], level_col[lv + 1]):
                for i in (np.nonzero(lp[j + 1:, j])[0] + j + 1).tolist():
                    ents.append((i, j, np.nonzero(lp[i, :j] & lp[j, :j])[0].tolist()))
            ents.sort(key=lambda t: (-len(t[2]), t[1], t[0]))      # longest K-loop first (LPT), then by column
            for i, j, kl in ents:
                col_row.append(i)
                ent_col.append(j)
                tile_k += kl
                tile_ij += [(i, j)] * len(kl)
                tile_kptr.append(len(tile_k))
            level_ent.append(len(col_row))
        slot = {(j, j): j for j in range(nt)}
        for e, (i, j) in enumerate(zip(col_row, ent_col)):
            slot[(i, j)] = nt + e
        self.slot, self.nslots = slot, nt + len(col_row)
        row_ptr, row_tile = [0], []
        for i in range(nt):
            row_tile += np.nonzero(lp[i, :i])[0].tolist()
            row_ptr.append(len(row_tile))
        i32 = lambda a: np.ascontiguousarray(np.asarray(a if len(a) else [0], dtype=np.int32))  # noqa: E731
        self.tables = dict(
            col_ptr=i32(np.zeros(nt + 1)),   # (not used by the level kernels: entries are addressed absolutely, ent_col names the column)
            col_row=i32(col_row), tile_kptr=i32(tile_kptr), tile_k=i32(tile_k), diag_kptr=i32(diag_kptr), diag_k=i32(diag_k),
            row_ptr=i32(row_ptr), row_tile=i32(row_tile),
            tile_sa=i32([slot[(j, k)] for (i, j), k in zip(tile_ij, tile_k)]),
            tile_sb=i32([slot[(i, k)] for (i, j), k in zip(tile_ij, tile_k)]),
            diag_s=i32([slot[(j, k)] for j in range(nt) for k in diag_k[diag_kptr[j]:diag_kptr[j + 1]]]),
            row_slot=i32([slot[(i, k)] for i in range(nt) for k in row_tile[row_ptr[i]:row_ptr[i + 1]]]),
            ent_col=i32(ent_col), tile_valid=i32(counts * dof))
        self.level_col = np.ascontiguousarray(level_col, dtype=np.int32)
        self.level_ent = np.ascontiguousarray(np.asarray(level_ent, dtype=np.int32))
        dk = np.diff(np.asarray(diag_kptr))
        self.level_maxk = np.ascontiguousarray(np.array([dk[level_col[lv]:level_col[lv + 1]].max() for lv in range(nlev)], dtype=np.int32))
        # the same tables per TREE level (the solves' schedule: a tree level = one or two consecutive launch levels)
        first = np.searchsorted(level, self.tree_level_col[:-1])          # first launch level of every tree level
        bounds = np.concatenate([level[self.tree_level_col[:-1]], [nlev]]).astype(np.int64)
        self.tree_level_ent = np.ascontiguousarray(self.level_ent[bounds].astype(np.int32))
        self.tree_level_maxk = np.ascontiguousarray(np.array([self.level_maxk[bounds[t]:bounds[t + 1]].max() for t in range(self.tree_levels)],
                                                             dtype=np.int32))
        del first
        self.col_count = np.zeros(nt, dtype=np.int32)     # (host tables of the column-by-column schedule: not used)
        self.l_tiles = int(lp.sum())
        self.tile_products = len(tile_k) + len(diag_k) + self.l_tiles
        t3 = float(TILE) ** 3
        self.flops = 2 * t3 * len(tile_k) + t3 * len(diag_k) + t3 * len(col_row) + t3 / 3 * nt
        self.dense_tile_products = sum(j * (nt - j) + (nt - j) for j in range(nt))
        self.dense_flops = sum((nt - 1 - j) * (2 * t3 * j + t3) + t3 * j + t3 / 3 for j in range(nt))
        # vectors: padded position of every column of the linearization, and back (-1: padding)
        col = (TILE * tile_of + dof * slot_of)[:, None] + np.arange(dof)[None, :]
        self.pad_of_col = np.ascontiguousarray(col.reshape(-1).astype(np.int32))
        src = np.full(self.npad, -1, dtype=np.int32)
        src[self.pad_of_col] = np.arange(self.n, dtype=np.int32)
        self.col_of_pad = src
        self._dev: Dict[str, Any] = {}

    def piece_tables(self, hblocks) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """tile_ptr / piece_blk / piece_rc of thx_hblock_layout for the PADDED tiles: every block lies in exactly one tile
        (``hblocks``: compiler.HessianBlocks of the linearization -- the block ids are its)."""
        d, nt = self.dof, self.ntiles
        b = np.asarray(hblocks.blocks, dtype=np.int64)
        ti, tj = self.tile_of[b[:, 0]], self.tile_of[b[:, 1]]
        t = ti * (ti + 1) // 2 + tj
        order = np.argsort(t, kind="stable")
        tile_ptr = np.zeros(nt * (nt + 1) // 2 + 1, dtype=np.int64)
        np.add.at(tile_ptr, t + 1, 1)
        tile_ptr = np.cumsum(tile_ptr).astype(np.int32)
        r0, c0 = d * self.slot_of[b[:, 0]], d * self.slot_of[b[:, 1]]
        rc = ((r0 & 0xFFFF) << 16) | (c0 & 0xFFFF)
        return tile_ptr, np.ascontiguousarray(order.astype(np.int32)), np.ascontiguousarray(rc[order].astype(np.uint32).view(np.int32))

    def _device(self, device):
        key = str(device)
        if key not in self._dev:
            t = {k: torch.from_numpy(v).to(device) for k, v in self.tables.items()}
            c = _lib.TilePattern()
            c.ntiles, c.nslots = self.ntiles, self.nslots
            for k, v in t.items():
                if k not in ("ent_col", "tile_valid"):
                    setattr(c, k, v.data_ptr())
            c.col_count_host = self.col_count.ctypes.data
            c.col_head_host = None
            ls = _lib.LevelSchedule()
            ls.nlevels = self.nlevels
            ls.level_col_host, ls.level_ent_host = self.level_col.ctypes.data, self.level_ent.ctypes.data
            ls.level_maxk_host = self.level_maxk.ctypes.data
            ls.level_stream_host = self.level_stream.ctypes.data if self.two_streams else None
            ls.ent_col, ls.tile_valid = t["ent_col"].data_ptr(), t["tile_valid"].data_ptr()
            # the triangular solves walk TREE levels (a tree level's columns are contiguous: the stream groups only order them
            # inside it) -- one launch per tree level and direction, on the caller's stream
            lt = _lib.LevelSchedule()
            lt.nlevels = self.tree_levels
            lt.level_col_host = self.tree_level_col.ctypes.data
            lt.level_ent_host, lt.level_maxk_host = self.tree_level_ent.ctypes.data, self.tree_level_maxk.ctypes.data
            lt.level_stream_host = None
            lt.ent_col, lt.tile_valid = t["ent_col"].data_ptr(), t["tile_valid"].data_ptr()
            vec = dict(pad_of_col=torch.from_numpy(self.pad_of_col).to(device), col_of_pad=torch.from_numpy(self.col_of_pad).to(device))
            self._dev[key] = (c, ls, t, vec, lt)
        return self._dev[key]

    def c_struct(self, device) -> _lib.TilePattern:
        return self._device(device)[0]

    def c_levels(self, device) -> _lib.LevelSchedule:
        return self._device(device)[1]

    def c_solve_levels(self, device) -> _lib.LevelSchedule:
        return self._device(device)[4]

    def vec_maps(self, device):
        return self._device(device)[3]


def tile_pattern(structure, dof: int) -> TilePattern:
    """Symbolic tile factorisation of a pose-graph structure (theseus_amd.compiler.PoseGraphStructure)."""
    return TilePattern(structure.num_cols, structure.lower_block_pattern(), dof)


class _LevelLayout:
    """thx_hblock_layout for thx_chol_factor_levels: the block list of the linearization (compiler.HessianBlocks: ids, block size)
    with tile_ptr / piece_* built for the padded tiles of a LevelPattern."""

    def __init__(self, pattern: LevelPattern, hblocks, device):
        tile_ptr, piece_blk, piece_rc = pattern.piece_tables(hblocks)
        self.t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device)
                  for k, v in dict(tile_ptr=tile_ptr, piece_blk=piece_blk, piece_rc=piece_rc).items()}
        base = hblocks.on(device)
        c = _lib.HBlockLayout()
        c.nblocks, c.bd, c.nvars, c.ntiles = hblocks.nblocks, hblocks.bd, hblocks.nvars, pattern.ntiles
        c.diag_blk, c.inc_blk = base.t["diag_blk"].data_ptr(), base.t["inc_blk"].data_ptr()
        for k, v in self.t.items():
            setattr(c, k, v.data_ptr())
        c.max_tile_pieces = _lib.max_offdiag_tile_pieces(tile_ptr, pattern.ntiles)
        self.c, self._base = c, base


class HipSparseCholeskyCore(HipCholeskyCore):
    """HipCholeskyCore whose factorisation follows the tile pattern of its linearization."""

    levels = False   # True: LevelPattern + thx_chol_factor_levels / thx_chol_solve_levels (elimination-tree parallelism)

    def _sparse_init(self, packed_factor: Optional[bool] = None, tile_count: Optional[Sequence[int]] = None):
        self._core_init()
        lin = self.linearization
        compact = bool(getattr(lin, "_compact", False))
        self.levels = tile_count is not None
        if self.levels:
            # LEVEL SCHEDULE: the linearization's column order is a tile-level nested dissection (tile_nested_dissection), tile j
            # holds tile_count[j] whole variables.  The solver works in the PADDED order (ntiles * 128): g is gathered into it,
            # delta gathered back (thx_vec_gather); H is read from the block list through piece tables built for the padded tiles.
            if not compact:
                raise RuntimeError("the level-scheduled solver needs the block-compact Hessian (SE3 pose graphs on the HIP kernels)")
            self.pattern = LevelPattern(lin.packed.structure.lower_block_pattern(), lin.packed.dof, tile_count)
            if self.pattern.nvars != lin.packed.structure.num_poses:
                raise ValueError("tile_count does not cover the linearization's variables")
            self.packed_factor = True
            self._level_layouts: Dict[str, Any] = {}
            return
        self.pattern = tile_pattern(lin.packed.structure, lin.packed.dof)
        # TILE-PACKED factor: L holds only the tiles of the pattern, (B, nslots, 128, 128), instead of a dense (B, ld, ld) frame
        # (n = 12288, chain graph: 12 MB instead of 604 MB per problem in fp32 -- the batch sizes of the reference's sweep fit).
        # Goes with the block-compact Hessian (thx_chol_factor_hblocks: neither H nor L is a dense frame then).
        if packed_factor and not compact:
            raise RuntimeError("packed_factor=True needs the block-compact Hessian (SE3 pose graphs on the HIP kernels)")
        self.packed_factor = compact if packed_factor is None else bool(packed_factor)

    def _level_layout(self, device):
        """thx_hblock_layout of the linearization's block list with the piece tables of the PADDED tiles (one per device)."""
        key = str(device)
        if key not in self._level_layouts:
            self._level_layouts[key] = _LevelLayout(self.pattern, self.linearization.packed.structure.hessian_blocks(), device)
        return self._level_layouts[key]

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
            if self.levels:   # vectors of the padded order: y (forward-substituted right-hand side) and the working vector
                # (zero once: the kernels write the matrix rows only -- the padding entries must stay exact zeros, they meet the
                #  zero columns of L in the fused forward substitution's products)
                self._yp = torch.zeros(B, self.pattern.npad, dtype=g.dtype, device=g.device)
                self._xp = torch.zeros(B, self.pattern.npad, dtype=g.dtype, device=g.device)

    def dense_factor(self) -> torch.Tensor:
        """The factor as a dense (B, ld, ld) lower-triangular frame (tests / inspection; the solver never builds it)."""
        if not self.packed_factor:
            return self.L
        lin = self.linearization
        ld = self.pattern.npad if self.levels else lin.ld      # (level schedule: the frame of the PADDED order)
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
        if self.levels:
            lin = self.linearization
            dev = self.L.device
            if rhs is None:
                self.K.chol_factor_levels(self._level_layout(dev), lin.Hc, lam, ellipsoidal_damping, damping_eps, self.L, self.panels,
                                          self.info, self.pattern)
                return None
            # y = L^-1 rhs in the padded order, fused into the diagonal launches (a handle _substitute recognises: the backward
            # half needs no second gather)
            self.K.vec_gather(rhs.contiguous(), self._xp, self.pattern.vec_maps(dev)["col_of_pad"])
            self.K.chol_factor_levels(self._level_layout(dev), lin.Hc, lam, ellipsoidal_damping, damping_eps, self.L, self.panels,
                                      self.info, self.pattern, rhs=self._xp, y=self._yp)
            return self._yp
        self._factor_call(lam, ellipsoidal_damping, damping_eps, rhs, y, pattern=self.pattern)
        return y

    def _solve_levels(self, L, panels, rhs, x, backward_only: bool):
        maps = self.pattern.vec_maps(L.device)
        if rhs is self._yp:
            src = self._yp
        else:
            self.K.vec_gather(rhs.contiguous(), self._xp, maps["col_of_pad"])
            src = self._xp
        self.K.chol_solve_levels(L, panels, src, self._xp, self.pattern, which=1 if backward_only else 0)
        self.K.vec_gather(self._xp, x, maps["pad_of_col"])

    def solve_with_snapshot(self, snapshot, rhs: torch.Tensor) -> torch.Tensor:
        """(L L^T)^-1 rhs with a kept copy of a factor: the list-driven solves along the pattern (the copy has the solver's own
        layout -- tile-packed or dense frame -- which thx_chol_solve would misread)."""
        L, panels = snapshot
        rhs = rhs.contiguous()
        x = torch.empty_like(rhs)
        if self.levels:
            self._solve_levels(L, panels, rhs, x, backward_only=False)
            return x
        self.K.chol_solve_sparse(L, self.linearization.n, panels, rhs, x, self.pattern, backward_only=False)
        return x

    def _substitute(self, rhs, x, backward_only: bool):
        """The triangular solves over the non-zero tiles of L only (thx_chol_solve_sparse; no limit on n)."""
        if self.levels:
            return self._solve_levels(self.L, self.panels, rhs, x, backward_only)
        self.K.chol_solve_sparse(self.L, self.linearization.n, self.panels, rhs, x, self.pattern, backward_only=backward_only)


class HipSparseCholeskySolver(HipSparseCholeskyCore, LinearSolver):
    """``linear_solver_cls`` for large pose graphs (SE3 / SE2 / SO3): tile-sparse factorisation under a fill-reducing
    variable ordering (pass ``linearization_kwargs=dict(ordering=...)`` to impose another one).

    ``ordering``: "auto" (default) | "nd" | "rcm".  SE3 pose graphs get a tile-level NESTED DISSECTION and the level-scheduled
    factorisation / solves (elimination-tree parallelism inside a problem: thx_chol_factor_levels) unless the time model prefers
    the band; "rcm" = reverse Cuthill-McKee + the column-by-column schedule (rounds 3-5)."""

    def __init__(self, objective: Objective, linearization_cls: Optional[Type[Linearization]] = None,
                 linearization_kwargs: Optional[Dict[str, Any]] = None, check_singular: bool = False,
                 packed_factor: Optional[bool] = None, ordering: str = "auto", batch_hint: Optional[int] = None, **kwargs):
        linearization_cls = linearization_cls or HipLinearization
        if not (isinstance(linearization_cls, type) and issubclass(linearization_cls, HipLinearization)):
            raise RuntimeError(f"HipSparseCholeskySolver only works with theseus_amd.HipLinearization, but {linearization_cls} "
                               "was provided.")
        linearization_kwargs = dict(linearization_kwargs or {})
        tile_count = None
        self.ordering_info: Dict[str, Any] = dict(method="given")
        if linearization_kwargs.get("ordering") is None:
            kern = linearization_kwargs.get("kernels")
            levels_possible = (kern is None or hasattr(kern, "pg_assemble_blocks")) and linearization_kwargs.get("block_hessian") is not False \
                and os.environ.get("THX_DENSE_HESSIAN", "0") != "1" and packed_factor is not False
            method = os.environ.get("THX_SPARSE_ORDERING", ordering)
            linearization_kwargs["ordering"], tile_count, self.ordering_info = level_ordering(
                objective, method=method if levels_possible else "rcm", batch_hint=batch_hint)
        LinearSolver.__init__(self, objective, linearization_cls, linearization_kwargs)
        if tile_count is not None and not getattr(self.linearization, "_compact", False):
            # (the linearization did not take the block-compact path after all: the column-by-column schedule on an RCM order)
            linearization_kwargs["ordering"] = fill_reducing_ordering(objective)
            tile_count, self.ordering_info = None, dict(method="rcm")
            LinearSolver.__init__(self, objective, linearization_cls, linearization_kwargs)
        self._check_singular = check_singular
        self._sparse_init(packed_factor, tile_count)

    def solve(self, damping: Optional[Union[float, torch.Tensor]] = None, ellipsoidal_damping: bool = True,
              damping_eps: float = 1e-8, check_info: bool = True, **kwargs) -> torch.Tensor:
        return self._solve(damping, ellipsoidal_damping, damping_eps, check_info)
