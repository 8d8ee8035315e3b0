"""Generic objectives on theseus_amd's own API (BASELINE.json configs[0]: examples/simple_example.py -- an
``AutoDiffCostFunction`` on a ``Vector``): any cost function that can hand over its weighted Jacobian blocks and error
(``CostFunction.weighted_jacobians_error``, theseus/core/cost_function.py:107-122), optimisation variables Euclidean
(``Vector`` / ``Point2`` / ``Point3``: retraction = addition, theseus/geometry/vector.py:177-178).

What runs where: the cost functions are evaluated by torch (they are user code -- the reference does the same,
core/cost_function.py:203-393); ``H = A^T A`` and ``g = A^T b`` are formed from the blocks by ``thx_block_assemble`` without the
dense ``A`` of DenseLinearization (dense_linearization.py:29-62), the damped system is factorised / solved by the tiled
Cholesky, LM's accept test and the masked retraction are the same kernels the pose-graph path uses.  The state the LM loop
moves around is ONE (B, n) tensor; the variables' tensors are column views of it.

Implicit backward (nonlinear_least_squares.py:121-135,265-292): the grad-enabled last Gauss-Newton step keeps H outside
autograd; ``g`` is formed by torch from the differentiable blocks, the solve's backward is one solve with the cached factor.
``backward_mode="unroll"`` / ``"truncated"`` (:222-282): the differentiated iterations keep H IN the graph -- torch forms the
(small) normal equations from the blocks, the damped solve runs on the kernels as in every other iteration and is one autograd
node whose backward is a solve with a copy of that iteration's factor (``_UnrolledSolve``).
"""
import warnings
from typing import List, Optional

import torch

from .core import Objective, Variable
from .generic import BlockAssembler
from .kernels import default_kernels, round_up


def _is_euclidean(v) -> bool:
    return "Vector" in {c.__name__ for c in type(v).__mro__}


class PackedEuclidean:
    """The packed-state interface of the optimiser loop (theseus_amd/packed.py: PackedPoseGraph) for Euclidean variables."""

    group = "Euclidean"

    def __init__(self, objective: Objective, kernels=None, order=None):
        from .packed import UnsupportedObjective
        self.objective = objective
        self.K = kernels or default_kernels()
        self.order = tuple(order) if order is not None else tuple(objective.optim_vars.keys())
        if sorted(self.order) != sorted(objective.optim_vars.keys()):
            raise ValueError("the variable ordering must hold every optimisation variable of the objective exactly once")
        self.vars: List[Variable] = [objective.optim_vars[name] for name in self.order]
        for v in self.vars:
            if not _is_euclidean(v):
                raise UnsupportedObjective(
                    f"HIP backend, generic path: optimisation variables must be Euclidean (Vector / Point2 / Point3); got "
                    f"{type(v).__name__} ({v.name}).  There is no CPU/eager fallback.")
        self.costs = list(objective.cost_functions.values())
        for c in self.costs:
            if not hasattr(c, "jacobians"):
                raise UnsupportedObjective(f"HIP backend, generic path: {type(c).__name__} ({c.name}) does not implement "
                                           "jacobians().")
        self.cols, col = [], 0
        for v in self.vars:
            self.cols.append((col, v.dof()))
            col += v.dof()
        self.n, self.m = col, objective.dim()
        self.ld = round_up(self.n, 32)
        index = {v.name: k for k, v in enumerate(self.vars)}
        self.cost_vars = [[index[v.name] for v in c.optim_vars()] for c in self.costs]
        self.asm = BlockAssembler(self.cols, self.cost_vars, [c.dim() for c in self.costs])
        self.version = objective.current_version
        self._state: Optional[torch.Tensor] = None
        self._views = None          # what the variables' tensors are when they view the state
        self._vars_stale = False
        self._state_exposed = False
        self._blocks = None         # weighted blocks of the last assemble(): the launch reads these tensors

    # ---- state <-> variables -----------------------------------------------------------------------------------------
    def _tracked(self):
        yield from self.vars
        seen = set()
        for c in self.costs:
            aux = c.aux_vars() + c.weight.aux_vars()
            for a in aux:
                if id(a) not in seen:
                    seen.add(id(a))
                    yield a

    def _pack(self):
        obj = self.objective
        obj._resolve_batch_size()
        B = obj.batch_size
        parts = [v.tensor if v.tensor.shape[0] == B else v.tensor.expand(B, -1) for v in self.vars]
        with torch.no_grad():
            self._state = torch.cat(parts, dim=1).contiguous()
        self._repoint()

    def _repoint(self):
        with torch.set_grad_enabled(self._state.requires_grad):
            self._views = [self._state[:, c0:c0 + d] for c0, d in self.cols]
        for v, t in zip(self.vars, self._views):
            v._tensor = t
        self._vars_stale = False
        self._state_exposed = True

    def sync(self, force: bool = False, deep: bool = False):
        """Re-pack when somebody replaced a variable's tensor (``Objective.update``, ``Variable.update``) -- an identity test
        per variable: generic objectives have a handful of variables."""
        if self._state is None or force:
            self._pack()
            return
        if self._vars_stale:
            return   # the loop owns the state; the variables are re-pointed at its end
        if any(v.tensor is not t for v, t in zip(self.vars, self._views)) or self._state.shape[0] != self._batch_of_vars():
            self._pack()

    def _batch_of_vars(self):
        self.objective._resolve_batch_size()
        return self.objective.batch_size

    def privatize_state(self):
        if self._state_exposed:
            with torch.no_grad():
                self._state = self._state.detach().clone()
            self._state_exposed = False
            self._vars_stale = True

    def flush_variables(self):
        if self._vars_stale and self._state is not None:
            self._repoint()

    @property
    def state(self):
        return self._state

    @property
    def device(self):
        return self._state.device

    @property
    def batch(self):
        return self._state.shape[0]

    @property
    def optim_variables(self):
        return self.vars

    def alloc_state(self):
        return torch.empty_like(self._state)

    def clone_state(self):
        return self._state.clone()

    def swap_state(self, new, repoint: bool = False):
        old = self._state
        self._state = new
        if repoint:
            self._repoint()
        else:
            self._vars_stale = True
        return old

    @staticmethod
    def _rec(t):   # (B, n) -> the (N = 1, B, record) layout of thx_copy_where / thx_vec_retract
        return t.view(1, *t.shape)

    def keep_where(self, mask, out):
        self.K.copy_where(mask, self._rec(self._state), self._rec(out))

    def copy_where(self, mask, src, dst):
        self.K.copy_where(mask, self._rec(src), self._rec(dst))

    def solution_dict(self, state):
        return {v.name: state[:, c0:c0 + d].cpu() for v, (c0, d) in zip(self.vars, self.cols)}

    def history_dict(self, hist, dtype):
        """(K + 1, B, n) states -> name -> (B, dof, K + 1) on the host (nonlinear_optimizer.py:150-163)."""
        h = hist.to(dtype).cpu()
        return {v.name: h[:, :, c0:c0 + d].permute(1, 2, 0).contiguous() for v, (c0, d) in zip(self.vars, self.cols)}

    # ---- evaluation (torch: the cost functions are user code) ----------------------------------------------------------
    class _at:
        """Context: the variables' tensors view ``state`` while the cost functions are evaluated."""

        def __init__(self, packed, state):
            self.p, self.state = packed, state

        def __enter__(self):
            self.saved = [v.tensor for v in self.p.vars]
            for v, (c0, d) in zip(self.p.vars, self.p.cols):
                v._tensor = self.state[:, c0:c0 + d]

        def __exit__(self, *a):
            for v, t in zip(self.p.vars, self.saved):
                v._tensor = t

    def error_vector(self, state=None):
        self.sync()
        with self._at(self, self._state if state is None else state):
            B = self.batch
            return torch.cat([c.weighted_error().expand(B, -1) for c in self.costs], dim=1)

    def error_metric(self, state=None, out: Optional[torch.Tensor] = None, poses=None):
        err = (self.error_vector(state if state is not None else poses) ** 2).sum(dim=1) / 2
        if out is not None:
            out.copy_(err)
            return out
        return err

    def weighted_blocks(self):
        """[(Jacobian blocks (B|1, dim, dof) per optimisation variable of the cost), ...], [weighted error (B|1, dim), ...]"""
        self.sync()
        Js, es = [], []
        with self._at(self, self._state):
            for c in self.costs:
                jac, err = c.weighted_jacobians_error()
                Js.append([j.contiguous() for j in jac])
                es.append(err.contiguous())
        return Js, es

    def assemble(self, H: torch.Tensor, g: torch.Tensor, graph: bool = False):
        """H (lower blocks) and g from the weighted blocks; ``graph``: also return g as a differentiable function of whatever the
        blocks depend on (the implicit step), H stays outside autograd."""
        Js, es = self.weighted_blocks()
        Jd = [[j.detach() for j in J] for J in Js]
        ed = [e.detach() for e in es]
        self.asm.assemble(self.K, Jd, ed, H, g)
        self._blocks = (Jd, ed)
        if not graph:
            return None
        B = self.batch
        parts = [None] * len(self.vars)
        for c, (J, e) in enumerate(zip(Js, es)):
            for s, k in enumerate(self.cost_vars[c]):
                term = -(J[s].transpose(1, 2) @ e.unsqueeze(2)).squeeze(2)
                parts[k] = term if parts[k] is None else parts[k] + term
        return torch.cat([(p.expand(B, -1) if p is not None else torch.zeros(B, d, dtype=H.dtype, device=H.device))
                          for p, (_, d) in zip(parts, self.cols)], dim=1)

    def supports_block_hessian(self) -> bool:
        return False

    def jacobian_blocks(self):
        return self._blocks[0] if self._blocks is not None else self.weighted_blocks()[0]

    def jacobian_times(self, blocks, v: torch.Tensor) -> torch.Tensor:
        """A v (B, m) from the per-cost blocks (Dogleg / trust-region ratio: dense_linearization.py:73-74)."""
        out = torch.zeros(v.shape[0], self.m, dtype=v.dtype, device=v.device)
        r = 0
        for c, J in enumerate(blocks):
            d = self.costs[c].dim()
            for s, k in enumerate(self.cost_vars[c]):
                c0, dof = self.cols[k]
                out[:, r:r + d] += (J[s] @ v[:, c0:c0 + dof].unsqueeze(2)).squeeze(2)
            r += d
        return out

    def dense_A_b(self):
        """Dense A (B, m, n), b (B, m) -- tests / foreign consumers only."""
        Js, es = self.weighted_blocks()
        B = self.batch
        A = torch.zeros(B, self.m, self.n, dtype=self._state.dtype, device=self._state.device)
        b = torch.zeros(B, self.m, dtype=self._state.dtype, device=self._state.device)
        r = 0
        for c, (J, e) in enumerate(zip(Js, es)):
            d = self.costs[c].dim()
            for s, k in enumerate(self.cost_vars[c]):
                c0, dof = self.cols[k]
                A[:, r:r + d, c0:c0 + dof] = J[s]
            b[:, r:r + d] = -e
            r += d
        return A, b

    def retract(self, delta: torch.Tensor, step: float, ignore_mask: Optional[torch.Tensor], out: torch.Tensor):
        """out = state + step * delta (rows of ``ignore_mask`` keep the state): thx_vec_retract on the one (B, n) record."""
        self.sync()
        m = None
        if ignore_mask is not None:
            m = ignore_mask if ignore_mask.dtype == torch.uint8 else ignore_mask.to(torch.uint8)
        self.K.vec_retract(self._rec(self._state), delta, 0, step, m, self._rec(out))
        return out

    # ---- BackwardMode.UNROLL / TRUNCATED: one iteration as a differentiable function of the state and of whatever the cost
    #      functions depend on (nonlinear_least_squares.py:100-215 with the Hessian IN the graph) -------------------------------
    def normal_equations(self, Js, es):
        """(H (B, n, n) symmetric, g (B, n)) from the weighted blocks, by torch: the differentiable twin of ``assemble``."""
        B = self.batch
        ref = es[0]
        H = torch.zeros(B, self.n, self.n, dtype=ref.dtype, device=ref.device)
        g = torch.zeros(B, self.n, dtype=ref.dtype, device=ref.device)
        for c, (J, e) in enumerate(zip(Js, es)):
            for sa, ka in enumerate(self.cost_vars[c]):
                ca, da = self.cols[ka]
                g[:, ca:ca + da] = g[:, ca:ca + da] - (J[sa].transpose(1, 2) @ e.unsqueeze(2)).squeeze(2)
                for sb, kb in enumerate(self.cost_vars[c]):
                    cb, db = self.cols[kb]
                    H[:, ca:ca + da, cb:cb + db] = H[:, ca:ca + da, cb:cb + db] + J[sa].transpose(1, 2) @ J[sb]
        return H, g

    def prepare_unroll(self):
        pass   # (the blocks are evaluated by torch at every differentiated iteration: nothing to re-pack)

    def where_state(self, mask: torch.Tensor, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return torch.where(mask.view(-1, 1), a, b)

    def state_with_graph(self, tensors):
        """The state assembled from the optimisation variables' OWN tensors (in ``optim_variables`` order), autograd history kept:
        where BackwardMode.UNROLL starts, so that gradients reach the values the caller passed in."""
        B = self._state.shape[0]
        return torch.cat([t if t.shape[0] == B else t.expand(B, -1) for t in tensors], dim=1)

    def unrolled_step(self, opt, X: torch.Tensor, frozen: Optional[torch.Tensor], kwargs):
        """X -> (X + step * delta where not ``frozen``, delta): the blocks are evaluated at X with the graph, the kernels get
        their detached values (what ``compute_delta`` factorises and LM's accept test reads), the solve is the autograd node."""
        solver = opt.linear_solver
        lin = solver.linearization
        lin._ensure_buffers()
        Js, es = [], []
        with self._at(self, X):
            for c in self.costs:
                jac, err = c.weighted_jacobians_error()
                Js.append([j.contiguous() for j in jac])
                es.append(err.contiguous())
        Jd, ed = [[j.detach() for j in J] for J in Js], [e.detach() for e in es]
        self.asm.assemble(self.K, Jd, ed, lin._H, lin.g)
        self._blocks = (Jd, ed)
        lin._after_assemble()
        H, g = self.normal_equations(Js, es)
        delta = _UnrolledSolve.apply(opt, kwargs, H, g)
        X_new = X + float(opt.params.step_size) * delta
        if frozen is not None:
            X_new = torch.where(frozen.view(-1, 1), X, X_new)
        return X_new, delta

    # ---- BackwardMode.IMPLICIT ----------------------------------------------------------------------------------------------
    def implicit_step(self, opt, step: float, kwargs):
        solver = opt.linear_solver
        lin = solver.linearization
        self.flush_variables()
        lin._ensure_buffers()
        g_graph = self.assemble(lin._H, lin.g, graph=True)
        lin._after_assemble()
        X = self._state.detach()
        delta = _CachedFactorSolve.apply(opt, kwargs, g_graph)
        return X + float(step) * delta, delta


class _CachedFactorSolve(torch.autograd.Function):
    """delta = H^-1 g with H outside autograd: forward = undamped factorisation (the reference falls back to the optimizer's
    damped step when it fails, nonlinear_least_squares.py:130-135) + the two substitutions; backward = one solve with the
    cached factor (the "backward linear solve" of the implicit mode)."""

    @staticmethod
    def forward(ctx, opt, kwargs, g):
        solver = opt.linear_solver
        lin = solver.linearization
        lin.g.copy_(g.detach())
        y = solver.factorize(None, rhs=lin.g)
        if bool(solver.info.ne(0).any()):
            if kwargs.get("__strict_implicit_final_gn__", False):
                solver.check_info()
            warnings.warn("implicit backward: the undamped Gauss-Newton system is not positive definite, "
                          "falling back to the optimizer's damped step", RuntimeWarning)
            delta = opt.compute_delta(**kwargs)
            solver.check_info()
        else:
            delta = torch.empty_like(lin.g)   # (NOT like y: the level schedule's y is its padded vector)
            solver._substitute(y, delta, backward_only=True)
        ctx.solver, ctx.factor_version = solver, solver.factor_version
        return delta

    @staticmethod
    def backward(ctx, grad_delta):
        solver = ctx.solver
        if solver.factor_version != ctx.factor_version:
            raise RuntimeError("implicit backward: the cached Cholesky factor of this forward pass was overwritten by "
                               "a later factorisation on the same optimizer; call backward() before the next forward().")
        return None, None, solver.solve_with_factor(grad_delta.contiguous())


class _UnrolledSolve(torch.autograd.Function):
    """delta = (H + D)^-1 g as a differentiable function of H and g (D: the optimizer's damping of this iteration, a function of
    diag(H) when ellipsoidal -- dense_solver.py:38-64).  Forward: the optimizer's own ``compute_delta`` on the kernels (the
    buffers hold the detached H, g).  Backward, with w = (H + D)^-1 grad_delta from a COPY of this iteration's factor (later
    iterations overwrite the solver's):  grad_g = w,  grad_H = -w delta^T - diag(lambda w * delta) [ellipsoidal]."""

    @staticmethod
    def forward(ctx, opt, kwargs, H, g):
        solver = opt.linear_solver
        delta = opt.compute_delta(**kwargs)
        if bool(solver.info.ne(0).any()):
            try:
                solver.check_info()
            except RuntimeError as run_err:
                raise RuntimeError(f"There was an error while running the linear optimizer. Original error message: {run_err}. "
                                   "Backward pass will not work. To obtain the best solution seen before the error, run with "
                                   "torch.no_grad()") from None
        damped, ellipsoidal, _ = solver._factored_with
        ctx.solver, ctx.n = solver, solver.linearization.n
        ctx.factor = solver.factor_snapshot()
        ctx.dropped = solver.dropped_mask() if hasattr(solver, "dropped_mask") else None   # check_singular: zero step, zero gradient
        ctx.lam = solver._lam.clone() if (damped and ellipsoidal) else None
        ctx.save_for_backward(delta)
        return delta

    @staticmethod
    def backward(ctx, grad_delta):
        (delta,) = ctx.saved_tensors
        if ctx.dropped is not None:
            grad_delta = grad_delta.masked_fill(ctx.dropped.unsqueeze(1), 0.0)
        w = ctx.solver.solve_with_snapshot(ctx.factor, grad_delta)
        if ctx.dropped is not None:
            w = w.masked_fill(ctx.dropped.unsqueeze(1), 0.0)
        grad_H = -(w.unsqueeze(2) * delta.unsqueeze(1))
        if ctx.lam is not None:
            grad_H = grad_H - torch.diag_embed(ctx.lam.view(-1, 1) * w * delta)
        return None, None, grad_H, w
