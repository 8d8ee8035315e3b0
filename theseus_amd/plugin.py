"""Adapters that plug the HIP back end into the REAL ``theseus`` (the drop-in boundary of SURVEY.md §8b).

    import theseus as th
    import theseus_amd.plugin as thp
    opt = th.LevenbergMarquardt(objective, linear_solver_cls=thp.HipCholeskySolver,
                                linearization_cls=thp.HipLinearization, vectorize=True)
    layer = th.TheseusLayer(opt)

The reference's optimiser loop (nonlinear_least_squares.py:100-215) stays untouched and talks to the two ABCs
``Linearization`` (theseus/optimizer/linearization.py:16-87) and ``LinearSolver``
(theseus/optimizer/linear/linear_solver.py:15-37); everything behind them runs in libtheseus_hip.so:

* SE3 pose graphs (``th.Between`` / ``th.Difference`` with Scale/Diagonal cost weights) take the FUSED path: the
  cost functions are never evaluated by torch, ``thx_pg_assemble`` builds H and g from the packed variables;
* every other objective (``th.AutoDiffCostFunction``, user cost functions, any variable type) takes the GENERIC
  path: the reference evaluates its (vectorised) weighted Jacobians, ``thx_block_assemble`` turns the blocks into
  H and g without the dense ``A`` of ``DenseLinearization`` (dense_linearization.py:29-62).

Both feed ``thx_chol_factor_forward`` / ``thx_chol_solve_backward``.  Autograd: H is always built outside the graph
(= ``_detach_hessian=True``, what ``backward_mode="implicit"`` asks for); ``Atb`` and ``solve()`` are
differentiable -- the solve's backward is one ``thx_chol_solve`` with the cached factor, the fused ``Atb``'s backward
is ``thx_pg_vjp``, the generic ``Atb`` is formed by torch from the reference's own differentiable Jacobians.
``backward_mode="unroll"`` / ``"truncated"`` (the Hessian in the graph): the generic path keeps H in torch's graph, the fused
pose-graph / bundle-adjustment paths make ``solve()`` one autograd node over the packed state and auxiliary tensors
(``_FusedUnrolledSolve`` / ``_FusedUnrolledSchurSolve``).

``theseus`` is imported at module import: this file is only usable where the reference is installed (it is not on
the GPU box of this build; tests/test_plugin_reference.py exercises it in the container that has /root/reference).
"""
from typing import Any, Dict, List, Optional, Type, Union

import torch

import theseus as th
from theseus.optimizer import Linearization as _RefLinearization
from theseus.optimizer.linear import CholeskyDenseSolver as _RefCholeskyDenseSolver
from theseus.optimizer.linear import LinearSolver as _RefLinearSolver

from theseus.global_params import _THESEUS_GLOBAL_PARAMS as _REF_GLOBAL_PARAMS

from .generic import BlockAssembler
from .autograd import detached_tensors, pg_vjp_grads
from .kernels import default_kernels, round_up
from .linear_solver import HipCholeskyCore
from .linearization import HipLinearizationCore
from .packed import UnsupportedObjective
from . import kernels as _kernels

# The Taylor-switch thresholds and fast_approx_local_jacobians are the REFERENCE's from here on: every launch reads
# theseus' / torchlie's own global-parameter objects (torchlie/global_params.py:44-68, theseus/global_params.py:46-80), so
# torchlie.set_global_params(...) / theseus.set_global_params(...) made by the user reach the HIP kernels.
_kernels.use_reference_global_params(_REF_GLOBAL_PARAMS)


def _refuse_unrolled_av(lin):
    """Differentiating THROUGH the iterations on a fused path: ``solve()`` is the autograd node (Gauss-Newton / Levenberg-Marquardt
    need nothing else); ``Av`` -- read by Dogleg / TrustRegion steps (dogleg.py:66, trust_region.py:97) -- comes from kernels outside
    autograd, so a gradient through it would be silently incomplete."""
    if getattr(lin, "_unroll", None) is not None and torch.is_grad_enabled():
        raise NotImplementedError(
            "backward_mode='unroll' / 'truncated' with gradients on a fused pose-graph / bundle-adjustment path differentiates "
            "solve() only (GaussNewton, LevenbergMarquardt); Av() of this linearization is outside autograd (Dogleg / TrustRegion "
            "steps).  Use backward_mode='implicit', or the generic path.")


class _FusedAtb(torch.autograd.Function):
    """g = A^T b of a pose graph as a differentiable function of the packed auxiliary tensors: forward is
    ``thx_pg_assemble`` (which also refreshes H), backward is ``thx_pg_vjp``."""

    @staticmethod
    def forward(ctx, lin, meas, w_between, prior_target, w_prior, lr_between, lr_prior):
        HipLinearizationCore._assemble(lin)
        t = lin.packed.tensors
        ctx.lin = lin
        ctx.tensors = detached_tensors(t, t.poses, meas, w_between, prior_target, w_prior, lr_between, lr_prior)
        return lin.g.clone()

    @staticmethod
    def backward(ctx, grad_g):
        lin = ctx.lin
        return (None,) + pg_vjp_grads(lin.K, lin.packed, ctx.tensors, grad_g.contiguous())


class _CachedFactorSolve(torch.autograd.Function):
    """delta = (H + damping)^-1 g with H outside autograd: backward = one solve with the cached factor."""

    @staticmethod
    def forward(ctx, solver, damping, ellipsoidal, eps, g):
        y = solver.factorize(damping, ellipsoidal, eps, rhs=g.detach().contiguous())
        delta = y.new_empty(y.shape[0], solver.linearization.n)   # (NOT like y: the level schedule's y is its padded vector)
        solver._substitute(y, delta, backward_only=True)   # (tile-sparse solver: the list-driven solve)
        solver.check_info()
        ctx.solver, ctx.version = solver, solver.factor_version
        return delta

    @staticmethod
    def backward(ctx, grad_delta):
        s = ctx.solver
        if s.factor_version != ctx.version:
            raise RuntimeError("implicit backward: the cached Cholesky factor of this forward pass was overwritten by "
                               "a later factorisation on the same solver; call backward() before the next forward().")
        return None, None, None, None, s.solve_with_factor(grad_delta.contiguous())


class _UnrolledFactorSolve(torch.autograd.Function):
    """delta = (H + D)^-1 g as a differentiable function of H AND g (backward_mode "unroll" / "truncated" on the generic path:
    the reference keeps the Hessian in the graph there, nonlinear_least_squares.py:100-135).  Forward: the kernels on the detached
    values.  Backward, with w = (H + D)^-1 grad_delta from a COPY of this call's factor (the loop factorises again before the
    backward runs):  grad_g = w,  grad_H = -w delta^T - diag(lambda w * delta) [ellipsoidal: D = lambda diag(H) + eps]."""

    @staticmethod
    def forward(ctx, solver, damping, ellipsoidal, eps, H, g):
        y = solver.factorize(damping, ellipsoidal, eps, rhs=g.detach().contiguous())
        delta = y.new_empty(y.shape[0], solver.linearization.n)   # (NOT like y: the level schedule's y is its padded vector)
        solver._substitute(y, delta, backward_only=True)
        solver.check_info()
        ctx.solver, ctx.n = solver, solver.linearization.n
        ctx.factor = solver.factor_snapshot()
        ctx.lam = solver._lam.clone() if (damping is not None and ellipsoidal) else None
        ctx.save_for_backward(delta)
        return delta

    @staticmethod
    def backward(ctx, grad_delta):
        (delta,) = ctx.saved_tensors
        w = ctx.solver.solve_with_snapshot(ctx.factor, grad_delta)
        grad_H = -(w.unsqueeze(2) * delta.unsqueeze(1))
        if ctx.lam is not None:
            grad_H = grad_H - torch.diag_embed(ctx.lam.view(-1, 1) * w * delta)
        return None, None, None, None, grad_H, w


class _FusedUnrolledSolve(torch.autograd.Function):
    """delta = (H(X, theta) + lambda I)^-1 g(X, theta) of an SE3 pose graph as a differentiable function of the packed poses AND the
    auxiliary tensors -- ``backward_mode="unroll"`` / ``"truncated"`` through the REAL loop, which linearizes with
    ``_detach_hessian=False`` (nonlinear_least_squares.py:100-135).  Forward: the damped factorisation and the solves on the kernels
    (H, g were assembled by ``linearize()`` at the detached values).  Backward: w = one solve with a COPY of this call's factor,
    then ``thx_pg_unroll_vjp`` (include/theseus_hip.h; theseus_amd/autograd.py:PGUnrolledIteration is the same node plus the
    retraction, for theseus_amd's own loop).  The retraction and the error evaluation in between are the reference's own
    differentiable ops."""

    @staticmethod
    def forward(ctx, solver, damping, ellipsoidal, eps, poses, meas, w_between, prior_target, w_prior, lr_between, lr_prior):
        lin = solver.linearization
        packed = lin.packed
        # the no-grad path's own solve: the unfused-forward fallback for systems beyond the fused substitution's LDS plan and the
        # check_singular masking included, so that a problem behaves the same with and without gradients
        delta = solver._solve(damping, ellipsoidal, eps, check_info=True)
        ctx.dropped = solver.dropped_mask()   # check_singular: items whose step was set to zero -- zero gradient, like the reference
        ctx.ell = solver._lam.clone() if (damping is not None and ellipsoidal) else None   # (lambda diag(H) is in the graph)
        ctx.packed, ctx.n = packed, lin.n
        ctx.tensors = detached_tensors(packed.tensors, poses, meas, w_between, prior_target, w_prior, lr_between, lr_prior)
        # one copy of the factor per differentiated iteration (dense frame: B x ld x ld elements; the tile-sparse solver's packed
        # factor: its non-zero tiles) -- later iterations overwrite the solver's own
        ctx.solver, ctx.factor = solver, solver.factor_snapshot()
        ctx.delta = delta.clone()
        return delta

    @staticmethod
    def backward(ctx, grad_delta):
        packed, t, K = ctx.packed, ctx.tensors, ctx.packed.K
        P, B = t.poses.shape[:2]
        dt, dev = t.poses.dtype, t.poses.device
        if ctx.dropped is not None:   # (the dropped items' factor may be broken: neither their grad_delta nor their w goes on)
            grad_delta = grad_delta.masked_fill(ctx.dropped.unsqueeze(1), 0.0)
        w = ctx.solver.solve_with_snapshot(ctx.factor, grad_delta)
        if ctx.dropped is not None:
            w = w.masked_fill(ctx.dropped.unsqueeze(1), 0.0)
        st = packed.structure
        E, Kp = st.num_edges, st.num_priors
        new = lambda *sh: torch.zeros(*sh, dtype=dt, device=dev)  # noqa: E731
        rec, dof = tuple(t.poses.shape[2:]), packed.dof
        gpi, gpj, gm, gwb = new(max(E, 1), B, *rec), new(max(E, 1), B, *rec), new(max(E, 1), B, *rec), new(max(E, 1), B, dof)
        gpp, gt, gwp = new(max(Kp, 1), B, *rec), new(max(Kp, 1), B, *rec), new(max(Kp, 1), B, dof)
        glb = new(max(E, 1), B, 1) if t.robust_between else None
        glp = new(max(Kp, 1), B, 1) if t.robust_prior else None
        K.pg_unroll_vjp(packed.dstruct, t, w, ctx.delta, gpi, gpj, gm, gwb, gpp, gt, gwp, ell_damping=ctx.ell, g_lrb=glb, g_lrp=glp)
        GX = torch.cat([gpi[:E], gpj[:E], gpp[:Kp], new(1, B, *rec)], 0)[packed.unroll_incidence(dev)].sum(1)

        def fit(g, count, like):
            if g is None or like is None:
                return None
            g = g[:count]
            return g.sum(1, keepdim=True) if like.shape[1] == 1 and B != 1 else g
        return (None, None, None, None, GX, fit(gm, E, t.meas), fit(gwb, E, t.w_between), fit(gt, Kp, t.prior_target),
                fit(gwp, Kp, t.w_prior), fit(glb, E, t.log_radius_between), fit(glp, Kp, t.log_radius_prior))


class _HipRetract(torch.autograd.Function):
    """X_new = X exp(delta) on the packed pose buffer (thx_se3_retract / thx_se2_retract), differentiable w.r.t. delta:
    the backward is thx_se3_retract_vjp / thx_se2_retract_vjp.  (X itself is the detached iterate of the no-grad loop.)"""

    @staticmethod
    def forward(ctx, packed, poses, delta, mask):
        out = torch.empty_like(poses)
        d = delta.detach().contiguous()
        packed.K.retract(poses, d, 1.0, mask, out)
        ctx.packed, ctx.mask = packed, mask
        ctx.save_for_backward(poses, d)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        poses, d = ctx.saved_tensors
        gd = torch.empty_like(d)
        ctx.packed.K.retract_vjp(poses, d, 1.0, grad_out.contiguous(), gd)
        if ctx.mask is not None:
            gd = gd * (ctx.mask == 0).to(gd.dtype).unsqueeze(1)
        return None, None, gd, None


class _FusedWeightedError:
    """What ``Objective.error()`` concatenates (core/objective.py:587-613): ONE item whose ``weighted_error()`` is the whole
    (B, m) weighted error vector in cost add order, from thx_pg_jacobians (residuals only)."""

    def __init__(self, packed):
        self.packed = packed

    def weighted_error(self) -> torch.Tensor:
        return self.packed.error_vector()


class _LazyJacobians:
    """``Objective._vectorized_jacobians_iter``: served on demand (``Objective._get_jacobians_iter``, objective.py:836-843).
    The fused linearization never asks; anybody else (another Linearization on the same objective, ``.A`` / ``.b``) gets the
    reference's own wrappers filled from thx_pg_jacobians -- or, when autograd history is needed, from the reference's
    differentiable vectorization."""

    def __init__(self, hooks):
        self.hooks = hooks

    def __iter__(self):
        h = self.hooks
        if h.needs_graph():
            h.ref["run"]()
            return iter(h.ref["jac_iter"])
        J0, J1, eb, Jp, ep = h.packed.jacobian_blocks()
        for w, (kind, k) in zip(h.ref["jac_iter"], h.slots):
            if kind == "e":
                w._cached_jacobians, w._cached_error = [J0[k], J1[k]], eb[k]
            else:
                w._cached_jacobians, w._cached_error = [Jp[k]], ep[k]
        return iter(h.ref["jac_iter"])


class HipObjectiveHooks:
    """The THIRD hook set of the boundary: the six swappable vectorization callbacks of ``th.Objective``
    (core/objective.py:128-138, installed like ``_enable_vectorization`` :916-943 does, undone by ``disable_vectorization``
    :945-951), called by the reference's loop from ``retract_vars_sequence`` / ``error_metric`` / ``update``
    (nonlinear_least_squares.py:315-325,361-364).  With them one iteration of the REAL ``th.LevenbergMarquardt`` on a
    pose graph is: thx_pg_assemble, thx_chol_factor_forward + thx_chol_solve_backward, thx_se3_retract, thx_pg_jacobians
    (residuals) -- no per-update torch Jacobian pass (``_vectorization_run`` becomes a no-op: the fused linearization reads
    the variables when it linearizes), no ATen retract, no ATen error evaluation.

    The reference's own ``Vectorize`` stays underneath: evaluations that must record autograd history (grad enabled and
    some tensor requires grad -- e.g. a user differentiating ``objective.error()``) are its differentiable torch ops; the
    grad-enabled retraction of ``backward_mode="implicit"``'s last step is the HIP kernel with its VJP kernel."""

    def __init__(self, objective, packed):
        from theseus.core.vectorizer import Vectorize
        if not objective.vectorized:
            Vectorize(objective)
        self.objective, self.packed = objective, packed
        self.ref = dict(jac_iter=objective._vectorized_jacobians_iter, run=objective._vectorization_run,
                        to=objective._vectorization_to, retract=objective._retract_method,
                        err_iter=objective._get_error_iter)
        self.pose_names = [v.name for v in packed.pose_vars]
        index_e = {id(c): k for k, c in enumerate(packed.edge_costs)}
        index_p = {id(c): k for k, c in enumerate(packed.prior_costs)}
        self.slots = []
        for wrapped in objective.cost_functions.values():
            c = getattr(wrapped, "cost_function", wrapped) if type(wrapped).__name__ == "RobustCostFunction" else wrapped
            self.slots.append(("e", index_e[id(c)]) if id(c) in index_e else ("p", index_p[id(c)]))
        objective._vectorized_jacobians_iter = _LazyJacobians(self)
        objective._vectorization_run = self.run
        objective._vectorization_to = self.to
        objective._retract_method = self.retract
        objective._get_error_iter = self.error_iter
        objective._vectorized = True
        objective._thx_hooks = self

    @staticmethod
    def install(objective, packed):
        h = getattr(objective, "_thx_hooks", None)
        if (isinstance(h, HipObjectiveHooks) and h.packed is packed and objective._vectorized
                and objective._retract_method == h.retract):
            return h
        return HipObjectiveHooks(objective, packed)

    def needs_graph(self) -> bool:
        return torch.is_grad_enabled() and any(v.tensor.requires_grad for v in self.packed.tracked_list())

    # ---- _vectorization_run: nothing to precompute (the reference's version is a full torch Jacobian pass per update) ----
    def run(self, *args, **kwargs):
        return None

    def to(self, *args, **kwargs):
        self.ref["to"](*args, **kwargs)
        self.packed.tensors = None      # re-packed on the new device / dtype at the next launch
        self.packed._known = []

    # ---- _retract_method (objective.py:873-914) ----
    def retract(self, delta, ordering, ignore_mask=None, force_update: bool = False):
        vars_ = list(ordering)
        packed = self.packed
        ts = [v.tensor for v in vars_]
        grad = torch.is_grad_enabled()
        if ([v.name for v in vars_] != self.pose_names or not delta.is_cuda and packed.K.name == "hip"
                or (grad and any(t.requires_grad for t in ts))):
            # not the packed pose order / differentiation w.r.t. the iterate itself: the reference's retraction
            return self.ref["retract"](delta, vars_, ignore_mask=ignore_mask, force_update=force_update)
        B = delta.shape[0]
        poses = packed.buffer_of(ts)
        if poses is None:
            poses = packed._stack([t.expand(B, *packed.gshape) if t.shape[0] != B else t for t in ts], B)
        mask = None
        if ignore_mask is not None and not force_update:
            mask = ignore_mask.to(torch.uint8).contiguous()
        if grad and delta.requires_grad:
            out = _HipRetract.apply(packed, poses.detach(), delta, mask)
        else:
            out = torch.empty_like(poses)
            packed.K.retract(poses, delta.detach().contiguous(), 1.0, mask, out)
        with torch.set_grad_enabled(grad and out.requires_grad):
            views = out.unbind(0)
        packed.remember_views(out, views)
        for v, t in zip(vars_, views):
            v.update(t)

    # ---- _get_error_iter (objective.py:587-613) ----
    def error_iter(self):
        if self.needs_graph():
            return self.ref["err_iter"]()
        return iter([_FusedWeightedError(self.packed)])


class HipLinearization(HipLinearizationCore, _RefLinearization):
    """Replaces ``th.DenseLinearization`` (theseus/optimizer/dense_linearization.py:15-77).  ``objective_hooks=False``
    keeps the reference's own vectorization callbacks (only ``Linearization`` + ``LinearSolver`` are replaced then)."""

    def __init__(self, objective: th.Objective, ordering=None, kernels=None, objective_hooks: bool = True,
                 block_hessian: Optional[bool] = None, **kwargs):
        self._ext_AtA = self._ext_Atb = None
        _RefLinearization.__init__(self, objective, ordering)
        self._g_graph: Optional[torch.Tensor] = None
        try:
            if not objective.cost_functions:
                raise UnsupportedObjective("an objective without cost functions")
            self._core_init(objective, kernels, block_hessian)
            self.fused = True
        except UnsupportedObjective:
            self.fused = False
            self._generic_init(objective, kernels)
        self.hooks = HipObjectiveHooks.install(objective, self.packed) if (self.fused and objective_hooks) else None

    # ---- generic path ------------------------------------------------------------------------------------
    def _generic_init(self, objective, kernels):
        self.K = kernels or default_kernels()
        self.packed = None
        self.asm, self._asm_sig = None, None   # compiled at the first linearize(): the column spans come from the Jacobians
        self._n, self._ld = self.num_cols, round_up(self.num_cols, 32)
        self._H_graph, self._detach_hessian_now = None, True
        self.H = self.g = None
        self._AtA_cache = self._A = self._b = None
        # column -> (position in self.ordering, offset inside that variable): any caller-supplied ordering (linearization.py:18-41)
        self._col_var = [(k, j) for k, d in enumerate(self.var_dims) for j in range(d)]

    def _column_blocks(self, cf, jacobians):
        """The reference writes Jacobian i of a cost function into the columns that START at the cost's i-th optimisation variable
        and span ``J.shape[2]`` of them (dense_linearization.py:44-52) -- a Jacobian may be wider than its variable (one (B, d, m)
        block for m one-dimensional variables that are adjacent in the ordering: tests/theseus_tests/optimizer/nonlinear/
        common.py:76-79), the list may be shorter than the variable list, and a later block overwrites an earlier one.  Returns
        [(variable position in the ordering, [(jacobian index, first column of it) | None per column of the variable])] for the
        variables the cost touches, ascending."""
        owner = {}
        for i, J in enumerate(jacobians):
            c0 = self.var_start_cols[self.ordering.index_of(cf.optim_var_at(i).name)]
            if J.ndim != 3 or c0 + J.shape[2] > self.num_cols:
                raise ValueError(f"cost function {cf.name}: Jacobian {i} of shape {tuple(J.shape)} does not fit into the columns "
                                 f"[{c0}, {self.num_cols}) it starts at")
            for j in range(J.shape[2]):
                owner[c0 + j] = (i, j)
        touched = {}
        for col, src in owner.items():
            k, off = self._col_var[col]
            touched.setdefault(k, [None] * self.var_dims[k])[off] = src
        return sorted(touched.items())

    @staticmethod
    def _gather_block(jacobians, cols):
        """One (B, dim, dof) block of a variable from its column sources (a plain view in the ordinary one-Jacobian-per-variable case)."""
        i0, j0 = cols[0] if cols[0] is not None else (None, None)
        if i0 is not None and all(c == (i0, j0 + t) for t, c in enumerate(cols)):
            J = jacobians[i0]
            return J if (j0 == 0 and J.shape[2] == len(cols)) else J[:, :, j0:j0 + len(cols)]
        ref = jacobians[next(c[0] for c in cols if c is not None)]
        B = max(J.shape[0] for J in jacobians)
        zero = ref.new_zeros(B, ref.shape[1], 1)
        return torch.cat([zero if c is None else jacobians[c[0]][:, :, c[1]:c[1] + 1].expand(B, -1, -1) for c in cols], dim=2)

    @property
    def n(self):
        return self.packed.n if self.fused else self._n

    @property
    def ld(self):
        return self.packed.ld if self.fused else self._ld

    def _weighted_blocks(self):
        """Per cost function: the weighted Jacobian block of every variable it touches + the weighted error; (re)compiles the block
        structure when the Jacobians' column spans are seen for the first time (or change)."""
        raw, sig = [], []
        for cf in self.objective._get_jacobians_iter():  # vectorised when the objective is (objective.py:836-843)
            jac, err = cf.weighted_jacobians_error()
            blocks = self._column_blocks(cf, jac)
            raw.append((jac, err, blocks, cf.dim()))
            sig.append(tuple((k, tuple(cols)) for k, cols in blocks))
        sig = tuple(sig)
        if sig != self._asm_sig:
            self.asm = BlockAssembler(list(zip(self.var_start_cols, self.var_dims)), [[k for k, _ in b] for _, _, b, _ in raw],
                                      [d for _, _, _, d in raw])
            self._asm_sig = sig
        Js = [[self._gather_block(jac, cols).contiguous() for _, cols in blocks] for jac, _, blocks, _ in raw]
        es = [err.contiguous() for _, err, _, _ in raw]
        return Js, es

    def _assemble_generic(self):
        Js, es = self._weighted_blocks()
        B, dev, dt = self.objective.batch_size, self.objective.device, self.objective.dtype
        if self.H is None or self.H.shape[0] != B or self.H.device != dev or self.H.dtype != dt:
            self.H = torch.zeros(B, self.ld, self.ld, dtype=dt, device=dev)  # zero once: fixed block pattern
            self.g = torch.empty(B, self.n, dtype=dt, device=dev)
        need_graph = torch.is_grad_enabled() and any(t.requires_grad for J in Js for t in J) or \
            (torch.is_grad_enabled() and any(e.requires_grad for e in es))
        Jd = [[j.detach() for j in J] for J in Js]
        ed = [e.detach() for e in es]
        self.asm.assemble(self.K, Jd, ed, self.H, self.g, gradient=not need_graph)
        self._blocks = (Jd, ed)  # the launch reads these tensors: keep them alive
        self._g_graph = self._H_graph = None
        if need_graph and not self._detach_hessian_now:
            # backward_mode "unroll" / "truncated": the Hessian is part of the graph -- its (small) dense form by torch from the
            # same differentiable blocks; the solve's backward returns grad_H (_UnrolledFactorSolve)
            Hg = torch.zeros(B, self.n, self.n, dtype=dt, device=dev)
            for c, J in enumerate(Js):
                for sa, va in enumerate(self.asm.cost_vars[c]):
                    ca, da = self.asm.var_cols[va]
                    for sb, vb in enumerate(self.asm.cost_vars[c]):
                        cb, db = self.asm.var_cols[vb]
                        Hg[:, ca:ca + da, cb:cb + db] = Hg[:, ca:ca + da, cb:cb + db] + J[sa].transpose(1, 2) @ J[sb]
            self._H_graph = Hg
        if need_graph:  # Atb = A^T b from the reference's differentiable Jacobians, block by block (no dense A)
            parts: List[Optional[torch.Tensor]] = [None] * len(self.var_dims)
            for c, (J, e) in enumerate(zip(Js, es)):
                for s, v in enumerate(self.asm.cost_vars[c]):
                    term = -(J[s].transpose(1, 2) @ e.unsqueeze(2)).squeeze(2)
                    parts[v] = term if parts[v] is None else parts[v] + term
            zero = lambda d: torch.zeros(B, d, dtype=dt, device=dev)  # noqa: E731
            self._g_graph = torch.cat([(p.expand(B, -1) if p is not None else zero(d)) for p, d in zip(parts, self.var_dims)], 1)
            self.g = self._g_graph.detach().contiguous()
        self._AtA_cache = self._A = self._b = None

    def _materialize_generic_A_b(self):
        Jd, ed = self._blocks
        B = self.H.shape[0]
        A = torch.zeros(B, self.num_rows, self.num_cols, dtype=self.H.dtype, device=self.H.device)
        b = torch.zeros(B, self.num_rows, dtype=self.H.dtype, device=self.H.device)
        r = 0
        for c, (J, e) in enumerate(zip(Jd, ed)):
            d = self.asm.cost_dims[c]
            for s, v in enumerate(self.asm.cost_vars[c]):
                c0, dof = self.asm.var_cols[v]
                A[:, r:r + d, c0:c0 + dof] = J[s]
            b[:, r:r + d] = -e
            r += d
        self._A, self._b = A, b

    # ``_AtA`` / ``_Atb`` are plain attributes of the reference's DenseLinearization (dense_linearization.py:24-27,60-62) that its
    # tests assign to and then modify IN PLACE through the ``AtA`` property (test_dense_solver.py:24-104): an assigned tensor is
    # kept as it is and served by ``AtA`` / ``Atb`` / ``solve()`` until the next ``linearize()``.
    @property
    def _AtA(self):
        if self._ext_AtA is not None:
            return self._ext_AtA
        return self._full_AtA() if (self.fused and self.linearized or not self.fused and self.H is not None) else None

    @_AtA.setter
    def _AtA(self, value):
        self._ext_AtA = value

    @property
    def _Atb(self):
        if self._ext_Atb is not None:
            return self._ext_Atb
        g = self._g_graph if self._g_graph is not None else self.g
        return None if g is None else g.unsqueeze(2)

    @_Atb.setter
    def _Atb(self, value):
        self._ext_Atb = value

    # ---- the ABC ---------------------------------------------------------------------------------------------
    def _linearize_jacobian_impl(self):
        if self.fused:
            self._materialize_A_b()
        else:
            self._materialize_generic_A_b()

    def _materialize_A_b(self):  # .A / .b properties of the core
        if self.fused:
            HipLinearizationCore._materialize_A_b(self)
        else:
            self._materialize_generic_A_b()

    def _linearize_hessian_impl(self, _detach_hessian: bool = False):
        self._ext_AtA = self._ext_Atb = None
        if not self.fused:
            self._detach_hessian_now = bool(_detach_hessian)
            self._assemble_generic()
            return   # (generic path: with the Hessian in the graph when the reference asks for it)
        else:
            packed = self.packed
            graph = torch.is_grad_enabled() and any(v.tensor.requires_grad for v in packed.tracked_list())
            self._unroll = None
            if graph and not _detach_hessian:
                # backward_mode "unroll" / "truncated": the Hessian is part of the graph.  SE3 pose graphs: H and g are assembled
                # at the detached values, solve() becomes ONE autograd node over (poses, auxiliary tensors) -- _FusedUnrolledSolve
                packed.prepare_unroll()      # (re-packs poses and auxiliary tensors WITH history)
                t = packed.tensors
                self._g_graph = None
                self._assemble()
                self._unroll = (t.poses, t.meas, t.w_between, t.prior_target, t.w_prior, t.log_radius_between, t.log_radius_prior)
                return
            if graph:
                packed.sync(force=True)  # re-pack WITH the autograd history of the auxiliary variables
                t = packed.tensors
                self._g_graph = _FusedAtb.apply(self, t.meas, t.w_between, t.prior_target, t.w_prior, t.log_radius_between,
                                                t.log_radius_prior)
            else:
                self._g_graph = None
                self._assemble()

    def Av(self, v: torch.Tensor) -> torch.Tensor:
        if self.fused:
            _refuse_unrolled_av(self)
            return HipLinearizationCore.Av(self, v)
        Jd, _ = self._blocks   # generic path: the reference's weighted Jacobian blocks of this linearization
        out = torch.zeros(v.shape[0], self.num_rows, dtype=v.dtype, device=v.device)
        r = 0
        for c, J in enumerate(Jd):
            d = self.asm.cost_dims[c]
            for s, k in enumerate(self.asm.cost_vars[c]):
                c0, dof = self.asm.var_cols[k]
                out[:, r:r + d] += (J[s] @ v[:, c0:c0 + dof].unsqueeze(2)).squeeze(2)
            r += d
        return out

    def hessian_approx(self):
        return self._full_AtA()

    def _ata_impl(self) -> torch.Tensor:
        return self._ext_AtA if self._ext_AtA is not None else self._full_AtA()

    def _atb_impl(self) -> torch.Tensor:
        if self._ext_Atb is not None:
            return self._ext_Atb
        g = self._g_graph if self._g_graph is not None else self.g
        return g.unsqueeze(2)


class _TensorSystem:
    """``AtA`` / ``Atb`` handed over as plain tensors instead of the packed Hessian of a HipLinearization: a foreign
    ``Linearization`` put on the solver after construction (tests/theseus_tests/optimizer/nonlinear/common.py:213-269), or
    ``linearization._AtA`` / ``._Atb`` assigned directly (tests/theseus_tests/optimizer/linear/test_dense_solver.py:24-104).
    The tensors are copied into the (B, ld, ld) frame the Cholesky kernels read; the originals are never written."""
    _compact = False
    linearized = True

    def __init__(self, K):
        self.K, self.H, self.g, self.n, self.ld = K, None, None, 0, 0

    def load(self, AtA: torch.Tensor, Atb: torch.Tensor):
        if AtA.ndim != 3 or AtA.shape[1] != AtA.shape[2]:
            raise ValueError("Matrix must have a 3 dimensions, the first one being a batch dimension, and be square.")
        B, n = AtA.shape[0], AtA.shape[1]
        ld = round_up(max(n, 1), 32)
        if self.H is None or tuple(self.H.shape) != (B, ld, ld) or self.H.dtype != AtA.dtype or self.H.device != AtA.device:
            self.H = torch.zeros(B, ld, ld, dtype=AtA.dtype, device=AtA.device)
        self.H[:, :n, :n].copy_(AtA.detach())
        self.g = Atb.detach().reshape(B, n).contiguous()
        self.n, self.ld = n, ld

    def diagonal(self) -> torch.Tensor:
        d = torch.empty(self.g.shape[0], self.n, dtype=self.g.dtype, device=self.g.device)
        self.K.diag(self.H, self.n, d)
        return d


class _TensorSystemSolver(HipCholeskyCore):
    def __init__(self, K, check_singular: bool):
        self.linearization = _TensorSystem(K)
        self._check_singular = check_singular
        self._core_init()


class HipCholeskySolver(HipCholeskyCore, _RefCholeskyDenseSolver):
    """Replaces ``th.CholeskyDenseSolver`` (theseus/optimizer/linear/dense_solver.py:20-161).

    It subclasses the reference's class so that ``LevenbergMarquardt``'s isinstance whitelists for ellipsoidal /
    adaptive damping accept it (levenberg_marquardt.py:21-48,82-87), but calls ``LinearSolver.__init__`` directly:
    ``DenseSolver.__init__`` rejects every linearization class that is not literally ``DenseLinearization``
    (dense_solver.py:28-32)."""

    def __init__(self, objective: th.Objective, linearization_cls: Optional[Type[_RefLinearization]] = None,
                 linearization_kwargs: Optional[Dict[str, Any]] = None, check_singular: bool = False,
                 lagged_failure_check: bool = False, **kwargs):
        linearization_cls = linearization_cls or HipLinearization
        if not (isinstance(linearization_cls, type) and issubclass(linearization_cls, HipLinearization)):
            raise RuntimeError("HipCholeskySolver only works with theseus_amd.plugin.HipLinearization, "
                               f"but {linearization_cls} was provided.")
        _RefLinearSolver.__init__(self, objective, linearization_cls, linearization_kwargs)
        self._check_singular = check_singular
        self._core_init()
        self._lagged = bool(lagged_failure_check)
        self._lag = None

    # ``lagged_failure_check=True`` (off by default; ``linear_solver_kwargs=dict(lagged_failure_check=True)``): ``solve()`` does not
    # wait for the factorisation to learn whether it failed.  The default look at ``info`` is the ONE host synchronisation the
    # plugin adds to an iteration of the reference's loop, and it sits right behind the 43 ms factorisation: everything the loop
    # does on the host afterwards (_step: ~1500 Variable.update calls, retraction, error, ~4.5 ms at 256 poses) runs with the GPU
    # idle.  Lagged: a failed item's step is zeroed ON THE DEVICE (its variables stay where they were), the flag travels to pinned
    # host memory behind the solve, and the RuntimeError is raised by the NEXT solve() -- inside the loop's try block, after the
    # loop's own end-of-iteration synchronisation, so it is never late by more than one iteration: status FAIL as in the
    # reference (nonlinear_least_squares.py:138-152), with one more (unchanged-state) column in the error history.  What the flag
    # gives up: a failure in the LAST iteration of an optimize() -- including the case where the dropped step makes the loop's
    # relative-error test fire (an unchanged error reads as "converged") -- is only a RuntimeWarning at the next reset() / solve(),
    # and the status says CONVERGED / MAX_ITERATIONS.  Hence opt-in.
    def reset(self, **kwargs):
        if self._lagged and self._lag is not None and self._lag.seen():
            self._lag = None
            import warnings
            warnings.warn("HipCholeskySolver(lagged_failure_check=True): the last linear solve of the previous optimize() failed "
                          "(not positive definite); its step was dropped.", RuntimeWarning)
        self._lag = None

    def _lagged_solve(self, damping, ellipsoidal_damping, damping_eps) -> torch.Tensor:
        from .nonlinear import _LaggedFlag
        if self._lag is not None and self._lag.seen():
            self._lag = None
            raise RuntimeError("linalg.cholesky: the factorisation of the previous linear solve could not be completed because the "
                               "input is not positive-definite (HipCholeskySolver(lagged_failure_check=True): reported one "
                               "solve late; that step was dropped).")
        delta = self._solve(damping, ellipsoidal_damping, damping_eps, check_info=False)
        failed = self.info.ne(0)
        delta.masked_fill_(failed.unsqueeze(1), 0.0)
        if self._lag is None:
            self._lag = _LaggedFlag(delta.device)
        self._lag.post(failed.any())
        return delta

    def solve(self, damping: Optional[Union[float, torch.Tensor]] = None, ellipsoidal_damping: bool = True,
              damping_eps: float = 1e-8, **kwargs) -> torch.Tensor:
        # failure = RuntimeError, which the reference loop turns into FAIL status under no_grad
        # (nonlinear_least_squares.py:138-152)
        lin = self.linearization
        if not isinstance(lin, HipLinearization) or lin._ext_AtA is not None or lin._ext_Atb is not None:
            return self._solve_tensor_system(lin.AtA, lin.Atb, damping, ellipsoidal_damping, damping_eps)
        if getattr(lin, "_unroll", None) is not None and torch.is_grad_enabled():
            if damping is not None and isinstance(damping, torch.Tensor) and damping.ndim > 1:
                raise ValueError("Damping must be a float or a 1-D tensor.")
            return _FusedUnrolledSolve.apply(self, damping, ellipsoidal_damping, damping_eps, *lin._unroll)
        g = lin._g_graph
        if g is not None and torch.is_grad_enabled():
            if damping is not None and isinstance(damping, torch.Tensor) and damping.ndim > 1:
                raise ValueError("Damping must be a float or a 1-D tensor.")
            Hg = getattr(lin, "_H_graph", None)
            if Hg is not None:
                return _UnrolledFactorSolve.apply(self, damping, ellipsoidal_damping, damping_eps, Hg, g)
            return _CachedFactorSolve.apply(self, damping, ellipsoidal_damping, damping_eps, g)
        if self._lagged:
            return self._lagged_solve(damping, ellipsoidal_damping, damping_eps)
        return self._solve(damping, ellipsoidal_damping, damping_eps, check_info=True)

    def _solve_tensor_system(self, AtA, Atb, damping, ellipsoidal_damping, damping_eps) -> torch.Tensor:
        """``solve()`` on whatever ``linearization.AtA`` / ``.Atb`` return (dense_solver.py:84-123 reads nothing else): same
        kernels, the system copied into their frame first (``_TensorSystem``)."""
        sub = getattr(self, "_tensor_solver", None)
        if sub is None:
            sub = self._tensor_solver = _TensorSystemSolver(self.K, self._check_singular)
        sub.linearization.load(AtA, Atb)
        if torch.is_grad_enabled() and (AtA.requires_grad or Atb.requires_grad):
            return _UnrolledFactorSolve.apply(sub, damping, ellipsoidal_damping, damping_eps, AtA,
                                              Atb.reshape(AtA.shape[0], AtA.shape[1]))
        return sub._solve(damping, ellipsoidal_damping, damping_eps, check_info=True)

    def _solve_sytem(self, Atb: torch.Tensor, AtA: torch.Tensor) -> torch.Tensor:  # abstract in DenseSolver
        raise NotImplementedError("HipCholeskySolver.solve() factorises its linearization's packed Hessian")


# ---- large pose graphs (theseus_amd/sparse.py): tile-sparse Cholesky under a fill-reducing ordering ----------------------------
from .sparse import HipSparseCholeskyCore, fill_reducing_ordering  # noqa: E402


class HipSparseCholeskySolver(HipSparseCholeskyCore, _RefCholeskyDenseSolver):
    """``linear_solver_cls`` for the REAL theseus loop in BaspachoSparseSolver's role (baspacho_sparse_solver.py:23-148) on
    SE3 / SE2 / SO3 pose graphs.  The fill-reducing ordering is a ``th.optimizer.VariableOrdering`` given to the linearization:
    the reference loop retracts and reads ``delta`` through it (nonlinear_least_squares.py:97)."""

    def __init__(self, objective: th.Objective, linearization_cls: Optional[Type[_RefLinearization]] = None,
                 linearization_kwargs: Optional[Dict[str, Any]] = None, check_singular: bool = False, **kwargs):
        linearization_cls = linearization_cls or HipLinearization
        if not (isinstance(linearization_cls, type) and issubclass(linearization_cls, HipLinearization)):
            raise RuntimeError(f"HipSparseCholeskySolver only works with theseus_amd.plugin.HipLinearization, but "
                               f"{linearization_cls} was provided.")
        linearization_kwargs = dict(linearization_kwargs or {})
        if linearization_kwargs.get("ordering") is None:
            linearization_kwargs["ordering"] = fill_reducing_ordering(objective, th.optimizer.VariableOrdering)
        _RefLinearSolver.__init__(self, objective, linearization_cls, linearization_kwargs)
        if not self.linearization.fused:
            raise NotImplementedError("HipSparseCholeskySolver: the tile pattern is derived from a fused pose-graph linearization")
        self._check_singular = check_singular
        self._sparse_init()

    def solve(self, damping: Optional[Union[float, torch.Tensor]] = None, ellipsoidal_damping: bool = True,
              damping_eps: float = 1e-8, **kwargs) -> torch.Tensor:
        lin = self.linearization
        if getattr(lin, "_unroll", None) is not None and torch.is_grad_enabled():
            # backward_mode "unroll" / "truncated": the same node as the dense solver's; its backward solves along the tile pattern
            # with a copy of this call's (tile-packed) factor
            if damping is not None and isinstance(damping, torch.Tensor) and damping.ndim > 1:
                raise ValueError("Damping must be a float or a 1-D tensor.")
            return _FusedUnrolledSolve.apply(self, damping, ellipsoidal_damping, damping_eps, *lin._unroll)
        g = lin._g_graph
        if g is not None and torch.is_grad_enabled():
            if damping is not None and isinstance(damping, torch.Tensor) and damping.ndim > 1:
                raise ValueError("Damping must be a float or a 1-D tensor.")
            return _CachedFactorSolve.apply(self, damping, ellipsoidal_damping, damping_eps, g)
        return self._solve(damping, ellipsoidal_damping, damping_eps, check_info=True)

    def _solve_sytem(self, Atb: torch.Tensor, AtA: torch.Tensor) -> torch.Tensor:  # abstract in DenseSolver
        raise NotImplementedError("HipSparseCholeskySolver.solve() factorises its linearization's packed Hessian")


# ---- bundle adjustment (theseus_amd/ba.py): Schur-complement linearization / solver for the REAL theseus loop ----------
from .ba import (BAImplicitStep, HipSchurLinearizationCore, HipSchurSolverCore, ba_unroll_backward, ba_vjp_grads,  # noqa: E402
                 detached_ba_tensors)


class _FusedAtbBA(torch.autograd.Function):
    """g = A^T b of a bundle-adjustment objective as a differentiable function of the packed auxiliary tensors (features,
    weights, calibration, log_loss_radius, prior targets): forward is ``thx_ba_assemble`` (which also refreshes the block
    Hessian, outside autograd), backward is ``thx_ba_vjp``."""

    @staticmethod
    def forward(ctx, lin, *aux):
        HipSchurLinearizationCore._assemble(lin)
        packed = lin.packed
        t = packed.tensors
        n_aux = len(BAImplicitStep.NAMES)
        aux, cc_aux = aux[:n_aux], aux[n_aux:]     # (+ the camera-camera Between costs' measurements and weights)
        ctx.lin = lin
        ctx.tensors = detached_ba_tensors(t, t.cams.detach(), t.points.detach(), aux)
        ctx.cc_tensors = None
        if cc_aux:
            import dataclasses
            ctx.cc_tensors = dataclasses.replace(packed.cc_tensors, poses=t.cams.detach(), meas=cc_aux[0].detach(),
                                                 w_between=cc_aux[1].detach())
        return lin.g.clone()

    @staticmethod
    def backward(ctx, grad_g):
        lin = ctx.lin
        packed, K = lin.packed, lin.K
        w = grad_g.contiguous()
        grads = ba_vjp_grads(K, packed, ctx.tensors, w)
        if ctx.cc_tensors is not None:   # thx_pg_vjp over the camera columns of w (as BAImplicitStep.backward, theseus_amd/ba.py)
            ct, E, B = ctx.cc_tensors, len(packed.cc_costs), w.shape[0]
            new = lambda *sh: torch.empty(*sh, dtype=w.dtype, device=w.device)  # noqa: E731
            g_meas, g_wb = new(E, B, 3, 4), new(E, B, 6)
            K.pg_vjp(packed.cc_dstruct, ct, w[:, :packed.nc].contiguous(), g_meas, g_wb, new(1, B, 3, 4), new(1, B, 6))
            fit = lambda g_, like: g_.sum(1, keepdim=True) if like.shape[1] == 1 and B != 1 else g_  # noqa: E731
            grads = grads + (fit(g_meas, ct.meas), fit(g_wb, ct.w_between))
        return (None,) + grads


class _CachedSchurSolve(torch.autograd.Function):
    """delta = (H + damping)^-1 g by point elimination, H outside autograd: backward = one solve with the cached factor of the
    reduced camera system (``HipSchurSolverCore.solve_with_factor``)."""

    @staticmethod
    def forward(ctx, solver, damping, ellipsoidal, eps, g):
        delta = solver._solve(damping, ellipsoidal, eps, check_info=True).clone()
        ctx.solver, ctx.version = solver, solver.factor_version
        return delta

    @staticmethod
    def backward(ctx, grad_delta):
        solver = ctx.solver
        if solver.factor_version != ctx.version:
            raise RuntimeError("backward of HipSchurSolver.solve(): the cached factor was overwritten by a later solve; "
                               "call backward() before the next forward().")
        return None, None, None, None, solver.solve_with_factor(grad_delta.contiguous())


class _FusedUnrolledSchurSolve(torch.autograd.Function):
    """delta = (H(X, theta) + D)^-1 g(X, theta) of a bundle-adjustment objective as a differentiable function of the packed cameras,
    points AND auxiliary tensors -- ``backward_mode="unroll"`` / ``"truncated"`` through the REAL loop (which linearizes with
    ``_detach_hessian=False``, nonlinear_least_squares.py:100-135).  Forward: point elimination + the reduced system's factorisation
    on the kernels.  Backward: ``ba_unroll_backward`` (theseus_amd/ba.py: the call's Schur system rebuilt at the saved tensors, one
    solve, ``thx_ba_unroll_vjp``).  The retraction and the error evaluation in between are the reference's own differentiable ops."""

    @staticmethod
    def forward(ctx, solver, damping, ellipsoidal, eps, cams, pts, *aux):
        import dataclasses
        packed = solver.linearization.packed
        n_aux = len(BAImplicitStep.NAMES)
        aux, cc_aux = aux[:n_aux], aux[n_aux:]
        delta = solver._solve(damping, ellipsoidal, eps, check_info=True).clone()
        ctx.solver, ctx.factor_args = solver, solver._factor_args
        ctx.tensors = detached_ba_tensors(packed.tensors, cams.detach(), pts.detach(), aux)
        ctx.cc_tensors = None
        if cc_aux:
            ctx.cc_tensors = dataclasses.replace(packed.cc_tensors, poses=cams.detach(), meas=cc_aux[0].detach(),
                                                 w_between=cc_aux[1].detach())
        ctx.delta = delta.clone()
        return delta

    @staticmethod
    def backward(ctx, grad_delta):
        solver = ctx.solver
        GC, GP, grads = ba_unroll_backward(solver.linearization.packed, solver, ctx.tensors, ctx.cc_tensors, ctx.factor_args,
                                           ctx.delta, grad_delta)
        return (None, None, None, None, GC, GP) + grads


class HipSchurLinearization(HipSchurLinearizationCore, _RefLinearization):
    """``linearization_cls`` for bundle-adjustment objectives (SE3 cameras + Point3 points, th.eb.Reprojection optionally
    robust, th.Difference priors).  Sets ``ordering`` = cameras, then points -- the reference retracts and reads ``delta``
    through ``linearization.ordering`` (nonlinear_least_squares.py:97, objective.py:873-914).

    Under grad (the last step of ``backward_mode="implicit"``, examples/bundle_adjustment.py:184-215 learns
    ``log_loss_radius`` this way): ``Atb`` carries the graph (``_FusedAtbBA``), the block Hessian is built outside autograd --
    which is what the reference asks for in that step (``_detach_hessian=True``, dense_linearization.py:61)."""

    def __init__(self, objective: th.Objective, ordering=None, kernels=None, **kwargs):
        packed, ordering = self._schur_setup(objective, ordering, kernels, th.optimizer.VariableOrdering)
        _RefLinearization.__init__(self, objective, ordering)
        self._schur_init(packed)
        self._g_graph: Optional[torch.Tensor] = None

    def _linearize_hessian_impl(self, _detach_hessian: bool = False):
        packed = self.packed
        graph = torch.is_grad_enabled() and any(v.tensor.requires_grad for v in packed.tracked_list())
        self._unroll = None
        if graph and not _detach_hessian:
            # backward_mode "unroll" / "truncated": the Hessian is part of the graph -- blocks assembled at the detached values,
            # solve() becomes ONE autograd node over (cameras, points, auxiliary tensors): _FusedUnrolledSchurSolve
            packed.prepare_unroll()      # (re-packs the state and the auxiliary tensors WITH history)
            t = packed.tensors
            self._g_graph = None
            self._assemble()
            cc = (packed.cc_tensors.meas, packed.cc_tensors.w_between) if packed.cc_costs else ()
            self._unroll = (t.cams, t.points, t.feat, t.w_obs, t.focal, t.k1, t.k2, t.log_radius_obs, t.cam_prior_target,
                            t.w_cam_prior, t.pt_prior_target, t.w_pt_prior) + cc
            return
        if graph:
            packed.sync(force=True)   # re-pack WITH the autograd history of the auxiliary variables
            t = packed.tensors
            cc = (packed.cc_tensors.meas, packed.cc_tensors.w_between) if packed.cc_costs else ()
            self._g_graph = _FusedAtbBA.apply(self, t.feat, t.w_obs, t.focal, t.k1, t.k2, t.log_radius_obs,
                                              t.cam_prior_target, t.w_cam_prior, t.pt_prior_target, t.w_pt_prior, *cc)
        else:
            self._g_graph = None
            self._assemble()

    def _atb_impl(self) -> torch.Tensor:
        g = self._g_graph if self._g_graph is not None else self.g
        return g.unsqueeze(2)

    def Av(self, v: torch.Tensor) -> torch.Tensor:
        _refuse_unrolled_av(self)
        return HipSchurLinearizationCore.Av(self, v)


class HipSchurSolver(HipSchurSolverCore, _RefCholeskyDenseSolver):
    """``linear_solver_cls`` for bundle adjustment.  Subclasses the reference's CholeskyDenseSolver for LevenbergMarquardt's
    isinstance whitelists (levenberg_marquardt.py:21-48,82-87) but calls ``LinearSolver.__init__`` directly, like
    HipCholeskySolver above; ``solve`` returns delta in ``linearization.ordering`` (cameras, then points)."""

    def __init__(self, objective: th.Objective, linearization_cls=None, linearization_kwargs=None, check_singular: bool = False,
                 sparse_reduced_system: bool = True, **kwargs):
        linearization_cls = linearization_cls or HipSchurLinearization
        if not (isinstance(linearization_cls, type) and issubclass(linearization_cls, HipSchurLinearization)):
            raise RuntimeError(f"HipSchurSolver only works with theseus_amd.plugin.HipSchurLinearization, but "
                               f"{linearization_cls} was provided.")
        _RefLinearSolver.__init__(self, objective, linearization_cls, linearization_kwargs)
        self._check_singular = check_singular
        self._schur_solver_init(sparse_reduced_system)

    def solve(self, damping=None, ellipsoidal_damping: bool = True, damping_eps: float = 1e-8, **kwargs) -> torch.Tensor:
        g = self.linearization._g_graph
        unroll = getattr(self.linearization, "_unroll", None)
        if unroll is not None and torch.is_grad_enabled():
            if damping is not None and isinstance(damping, torch.Tensor) and damping.ndim > 1:
                raise ValueError("Damping must be a float or a 1-D tensor.")
            return _FusedUnrolledSchurSolve.apply(self, damping, ellipsoidal_damping, damping_eps, *unroll)
        if g is not None and torch.is_grad_enabled():
            if damping is not None and isinstance(damping, torch.Tensor) and damping.ndim > 1:
                raise ValueError("Damping must be a float or a 1-D tensor.")
            return _CachedSchurSolve.apply(self, damping, ellipsoidal_damping, damping_eps, g)
        return self._solve(damping, ellipsoidal_damping, damping_eps, check_info=True).clone()

    def _solve_sytem(self, Atb, AtA):  # abstract in DenseSolver
        raise NotImplementedError("HipSchurSolver.solve() eliminates the points of its block linearization")
