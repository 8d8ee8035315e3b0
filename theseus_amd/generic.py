"""Generic block assembly for objectives the fused pose-graph kernels do not cover.

``BlockAssembler`` turns "a list of cost functions, each with its (weighted) Jacobian blocks and error" -- what
``CostFunction.weighted_jacobians_error`` returns in the reference (theseus/core/cost_function.py:107-122) -- into the
lower triangle of H = A^T A and g = A^T b through ``thx_block_assemble`` (csrc/block_kernels.hip), never forming the
dense A of ``DenseLinearization`` (theseus/optimizer/dense_linearization.py:29-62).  The block structure (which
variable pairs meet in a cost) is compiled once; only the pointer tables are refreshed per call.
"""
from typing import List, Sequence, Tuple

import numpy as np
import torch

H_TARGET = np.dtype([("row0", "<i4"), ("col0", "<i4"), ("dof_a", "<i4"), ("dof_b", "<i4"), ("term_begin", "<i4"),
                     ("term_end", "<i4"), ("elem_begin", "<i4"), ("pad", "<i4")])
H_TERM = np.dtype([("Ja", "<u8"), ("Jb", "<u8"), ("bstride_a", "<i8"), ("bstride_b", "<i8"), ("dim", "<i4"),
                   ("dof_a", "<i4"), ("dof_b", "<i4"), ("pad", "<i4")])
G_TARGET = np.dtype([("col0", "<i4"), ("dof", "<i4"), ("term_begin", "<i4"), ("term_end", "<i4"), ("elem_begin", "<i4"),
                     ("pad", "<i4")])
G_TERM = np.dtype([("J", "<u8"), ("e", "<u8"), ("bstride_j", "<i8"), ("bstride_e", "<i8"), ("dim", "<i4"), ("dof", "<i4")])
assert (H_TARGET.itemsize, H_TERM.itemsize, G_TARGET.itemsize, G_TERM.itemsize) == (32, 48, 24, 40)


class BlockAssembler:
    def __init__(self, var_cols: Sequence[Tuple[int, int]], cost_vars: Sequence[Sequence[int]], cost_dims: Sequence[int]):
        """var_cols[v] = (first column, dof) of variable v in the linearization's ordering; cost_vars[c] = indices of
        the optimisation variables of cost c (Jacobian list order); cost_dims[c] = rows of cost c."""
        self.var_cols = [tuple(map(int, vc)) for vc in var_cols]
        self.cost_vars = [list(map(int, cv)) for cv in cost_vars]
        self.cost_dims = [int(d) for d in cost_dims]
        self.n = sum(d for _, d in self.var_cols)
        pairs = {}   # (va, vb) with col(va) >= col(vb)  ->  [(cost, slot_a, slot_b)]
        grads = {}   # v -> [(cost, slot)]
        for c, vs in enumerate(self.cost_vars):
            for sa, va in enumerate(vs):
                grads.setdefault(va, []).append((c, sa))
                for sb, vb in enumerate(vs):
                    ca, cb = self.var_cols[va][0], self.var_cols[vb][0]
                    if ca > cb or (ca == cb and sa == sb):
                        pairs.setdefault((va, vb), []).append((c, sa, sb))
                    elif ca == cb and sa != sb:
                        raise ValueError("a cost function lists the same optimisation variable twice")
        self.h_keys = sorted(pairs, key=lambda k: (self.var_cols[k[0]][0], self.var_cols[k[1]][0]))
        self.h_terms_of = [pairs[k] for k in self.h_keys]
        self.g_keys = sorted(grads, key=lambda v: self.var_cols[v][0])
        self.g_terms_of = [grads[v] for v in self.g_keys]
        ht = np.zeros(len(self.h_keys), H_TARGET)
        elems, e2t, tb = 0, [], 0
        for t, (va, vb) in enumerate(self.h_keys):
            (ra, da), (cb_, db) = self.var_cols[va], self.var_cols[vb]
            nt = len(self.h_terms_of[t])
            ht[t] = (ra, cb_, da, db, tb, tb + nt, elems, 0)
            e2t += [t] * (da * db)
            elems += da * db
            tb += nt
        self.h_targets, self.h_e2t, self.n_h_elems, self.n_h_terms = ht, np.asarray(e2t, np.int32), elems, tb
        gt = np.zeros(len(self.g_keys), G_TARGET)
        elems, e2t, tb = 0, [], 0
        for t, v in enumerate(self.g_keys):
            c0, d = self.var_cols[v]
            nt = len(self.g_terms_of[t])
            gt[t] = (c0, d, tb, tb + nt, elems, 0)
            e2t += [t] * d
            elems += d
            tb += nt
        self.g_targets, self.g_e2t, self.n_g_elems, self.n_g_terms = gt, np.asarray(e2t, np.int32), elems, tb
        self._dev = {}

    def lower_block_pattern(self):
        """[(row0, col0, dof_a, dof_b)] of the non-zero blocks of tril(H)."""
        return [(int(t["row0"]), int(t["col0"]), int(t["dof_a"]), int(t["dof_b"])) for t in self.h_targets]

    def _static(self, device):
        key = str(device)
        if key not in self._dev:
            up = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(device)  # noqa: E731
            self._dev[key] = (up(self.h_targets), up(self.h_e2t), up(self.g_targets), up(self.g_e2t))
        return self._dev[key]

    def term_tables(self, jacobians: List[List[torch.Tensor]], errors: List[torch.Tensor]):
        """Pointer tables for the current Jacobian / error tensors (contiguous (B|1, dim, dof) and (B|1, dim))."""
        ht = np.zeros(self.n_h_terms, H_TERM)
        k = 0
        for terms in self.h_terms_of:
            for (c, sa, sb) in terms:
                Ja, Jb = jacobians[c][sa], jacobians[c][sb]
                ht[k] = (Ja.data_ptr(), Jb.data_ptr(), Ja.stride(0) if Ja.shape[0] > 1 else 0,
                         Jb.stride(0) if Jb.shape[0] > 1 else 0, Ja.shape[1], Ja.shape[2], Jb.shape[2], 0)
                k += 1
        gt = np.zeros(self.n_g_terms, G_TERM)
        k = 0
        for terms in self.g_terms_of:
            for (c, sa) in terms:
                J, e = jacobians[c][sa], errors[c]
                gt[k] = (J.data_ptr(), e.data_ptr(), J.stride(0) if J.shape[0] > 1 else 0,
                         e.stride(0) if e.shape[0] > 1 else 0, J.shape[1], J.shape[2])
                k += 1
        return ht, gt

    def check(self, jacobians, errors):
        for c, (Js, e) in enumerate(zip(jacobians, errors)):
            if len(Js) != len(self.cost_vars[c]):
                raise ValueError(f"cost {c}: expected {len(self.cost_vars[c])} Jacobian blocks, got {len(Js)}")
            for s, J in enumerate(Js):
                dof = self.var_cols[self.cost_vars[c][s]][1]
                if J.ndim != 3 or J.shape[1] != self.cost_dims[c] or J.shape[2] != dof or not J.is_contiguous():
                    raise ValueError(f"cost {c} slot {s}: Jacobian must be contiguous (B, {self.cost_dims[c]}, {dof}), "
                                     f"got {tuple(J.shape)}")
            if e.ndim != 2 or e.shape[1] != self.cost_dims[c] or not e.is_contiguous():
                raise ValueError(f"cost {c}: error must be contiguous (B, {self.cost_dims[c]})")

    def assemble(self, K, jacobians, errors, H: torch.Tensor, g: torch.Tensor, hessian: bool = True, gradient: bool = True):
        """H[:, lower blocks] = sum J_a^T J_b ; g = -sum J^T e.  The tensors in ``jacobians`` / ``errors`` must stay
        alive until the launch has run (the caller's stream order guarantees that for ordinary use)."""
        self.check(jacobians, errors)
        K.block_assemble(self, jacobians, errors, H if hessian else None, g if gradient else None)
