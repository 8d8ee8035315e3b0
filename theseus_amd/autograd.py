"""Autograd glue of the implicit backward mode (theseus/optimizer/nonlinear/nonlinear_least_squares.py:121-135,
265-292).

Forward: one undamped Gauss-Newton step at the (detached) iterate of the no-grad loop,
``X_new = X exp(step * delta)``, ``delta = H^-1 g(theta)``, H outside autograd.  Backward: three launches --
``thx_se3_retract_vjp`` (grad_X_new -> grad_delta), ``thx_chol_solve`` with the factor cached by the forward
(the "backward linear solve"), ``thx_pg_vjp`` (grad w.r.t. measurements, prior targets and cost weights).
No torch ops compute anything here except the reductions over broadcast dimensions.
"""
import dataclasses
import warnings

import torch


def detached_tensors(t, poses, meas, w_between, prior_target, w_prior, lr_between, lr_prior):
    """PGTensors for a backward pass: same problem, no autograd history."""
    det = lambda x: None if x is None else x.detach()  # noqa: E731
    return dataclasses.replace(t, poses=poses.detach(), meas=det(meas), w_between=det(w_between),
                               prior_target=det(prior_target), w_prior=det(w_prior),
                               log_radius_between=det(lr_between), log_radius_prior=det(lr_prior))


def pg_vjp_grads(K, packed, t, w):
    """thx_pg_vjp + the reductions over broadcast batch dimensions: gradients of w^T g w.r.t. (measurements,
    between weights, prior targets, prior weights, log_loss_radius of the robust between / prior costs)."""
    B = t.poses.shape[1]
    E, Kp = packed.structure.num_edges, packed.structure.num_priors
    new = lambda *s: torch.empty(*s, dtype=w.dtype, device=w.device)  # noqa: E731
    gs, dof = t.poses.shape[2:], packed.dof   # group record shape (3,4) | (4,) | (3,3), tangent size 6 | 3 | 3
    g_meas, g_wb = new(max(E, 1), B, *gs), new(max(E, 1), B, dof)
    g_tgt, g_wp = new(max(Kp, 1), B, *gs), new(max(Kp, 1), B, dof)
    g_lrb = new(max(E, 1), B, 1) if t.robust_between else None
    g_lrp = new(max(Kp, 1), B, 1) if t.robust_prior else None
    K.pg_vjp(packed.dstruct, t, w, g_meas, g_wb, g_tgt, g_wp, g_lrb=g_lrb, g_lrp=g_lrp)

    def fit(g, count, like):  # (count, B, ...) -> the packed input's shape (count, 1|B, ...)
        if g is None or like is None:
            return None
        g = g[:count]
        return g.sum(1, keepdim=True) if like.shape[1] == 1 and B != 1 else g
    return (fit(g_meas, E, t.meas), fit(g_wb, E, t.w_between), fit(g_tgt, Kp, t.prior_target), fit(g_wp, Kp, t.w_prior),
            fit(g_lrb, E, t.log_radius_between), fit(g_lrp, Kp, t.log_radius_prior))


class ImplicitStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, opt, step, kwargs, meas, w_between, prior_target, w_prior, lr_between, lr_prior):
        solver = opt.linear_solver
        lin = solver.linearization
        packed = lin.packed
        lin._assemble()
        y = solver.factorize(None, rhs=lin.g)  # plain GN; the reference falls back to the damped step if it fails
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
        X = packed.tensors.poses.detach()
        X_new = torch.empty_like(X)
        packed.retract(delta, step, None, X_new)  # force_update: the converged mask is ignored in this step
        ctx.opt, ctx.step = opt, step
        ctx.factor_version = solver.factor_version
        ctx.tensors = detached_tensors(packed.tensors, X, meas, w_between, prior_target, w_prior, lr_between, lr_prior)
        ctx.delta = delta
        ctx.mark_non_differentiable(delta)
        return X_new, delta

    @staticmethod
    def backward(ctx, grad_x, _grad_delta):
        solver = ctx.opt.linear_solver
        lin = solver.linearization
        packed, K = lin.packed, solver.K
        if solver.factor_version != ctx.factor_version:
            raise RuntimeError("implicit backward: the cached Cholesky factor of this forward pass was overwritten by "
                               "a later factorisation on the same optimizer; call backward() before the next forward().")
        t = ctx.tensors
        B, n = t.poses.shape[1], lin.n
        gd = torch.empty(B, n, dtype=t.poses.dtype, device=t.poses.device)
        K.retract_vjp(t.poses, ctx.delta, ctx.step, grad_x.contiguous(), gd)
        w = solver.solve_with_factor(gd)  # the backward linear solve
        return (None, None, None) + pg_vjp_grads(K, packed, t, w)


def compose_left_backward(K, group: str, X, delta, step: float, G):
    """grad w.r.t. X of X_new = X exp(step * delta) given G = grad_X_new (raw entries), as the reference's autograd produces it:
    SE3 / SO3 -- torchlie's Compose.backward (se3_impl.py:739-747, so3_impl.py:702-707): the plain matrix rule
    [G_R E_R^T + G_t E_t^T | G_t] / G E^T; SE2 -- plain autograd through theseus/geometry/se2.py's compose on [x, y, cos, sin]."""
    P, B = X.shape[:2]
    dof = delta.shape[1] // P
    xi = (step * delta).view(B, P, dof).transpose(0, 1).reshape(P * B, dof).contiguous()
    if group == "SE3":
        E = K.se3_exp(xi).view(P, B, 3, 4)
        return torch.cat([G[..., :3] @ E[..., :3].transpose(-1, -2) + G[..., 3:] @ E[..., 3:].transpose(-1, -2), G[..., 3:]], -1)
    if group == "SO3":
        E = K.so3_exp(xi).view(P, B, 3, 3)
        return G @ E.transpose(-1, -2)
    E = K.se2_exp(xi).view(P, B, 4)
    ex, ey, ec, es = E.unbind(-1)
    gx, gy, gc, gs = G.unbind(-1)
    # x' = x + c ex - s ey,  y' = y + s ex + c ey,  c' = c ec - s es,  s' = s ec + c es
    return torch.stack([gx, gy, gx * ex + gy * ey + gc * ec + gs * es, -gx * ey + gy * ex - gc * es + gs * ec], -1)


class PGUnrolledIteration(torch.autograd.Function):
    """One DIFFERENTIATED iteration of an SE3 / SE2 / SO3 pose graph (BackwardMode.UNROLL / TRUNCATED,
    theseus/optimizer/nonlinear/nonlinear_least_squares.py:223-292: the Hessian is part of the graph):
    ``X_new = X exp(step * delta)``, ``delta = (H(X, theta) + lambda I)^-1 g(X, theta)``.  Forward: the optimizer's own kernels
    at the detached iterate (assemble, damped factorisation, solves, retraction).  Backward, given grad_X_new (raw 3 x 4 entries):
    thx_se3_retract_vjp -> grad_delta; Compose.backward's plain matrix rule -> the direct path to grad_X; one thx_chol_solve with a
    COPY of this iteration's factor -> w; thx_pg_unroll_vjp(w, delta) -> the per-cost gradients of phi = -(J w).(r + J delta)
    [- lambda sum_i w_i delta_i H_ii with ellipsoidal damping]
    w.r.t. both poses, the measurement / target and the weights (include/theseus_hip.h).  The poses' gradients are summed over
    each pose's incident costs in a fixed order (no atomics)."""

    @staticmethod
    def forward(ctx, opt, packed, frozen, kwargs, X, meas, w_between, prior_target, w_prior, lr_between, lr_prior):
        solver = opt.linear_solver
        lin, K = solver.linearization, packed.K
        Xd = X.detach().contiguous()
        t = packed.tensors
        t.poses = Xd                       # the kernels linearise at the tail loop's iterate
        lin._assemble()
        delta = opt.compute_delta(**kwargs)
        if bool(solver.info.ne(0).any()):
            try:
                solver.check_info()
            except RuntimeError as run_err:
                raise RuntimeError(f"There was an error while running the linear optimizer. Original error message: {run_err}. "
                                   "Backward pass will not work. To obtain the best solution seen before the error, run with "
                                   "torch.no_grad()") from None
        damped, ellipsoidal, _ = solver._factored_with
        ctx.dropped = solver.dropped_mask() if hasattr(solver, "dropped_mask") else None   # check_singular: zero step, zero gradient
        ctx.ell = solver._lam.clone() if (damped and ellipsoidal) else None   # D = lambda diag(H) + eps: lambda diag(H) is in the graph
        step = float(opt.params.step_size)
        mask = frozen.to(torch.uint8).contiguous() if frozen is not None else None
        X_new = torch.empty_like(Xd)
        K.retract(Xd, delta, step, mask, X_new)
        ctx.packed, ctx.step, ctx.frozen, ctx.n = packed, step, frozen, lin.n
        ctx.tensors = detached_tensors(t, Xd, meas, w_between, prior_target, w_prior, lr_between, lr_prior)
        ctx.solver, ctx.factor = solver, solver.factor_snapshot()   # (later iterations overwrite the solver's factor)
        ctx.delta = delta.detach().clone()
        ctx.mark_non_differentiable(delta)
        return X_new, delta

    @staticmethod
    def backward(ctx, G, _grad_delta):
        packed, t, K, n, step = ctx.packed, ctx.tensors, ctx.packed.K, ctx.n, ctx.step
        X, delta = t.poses, ctx.delta
        P, B = X.shape[:2]
        dt, dev = X.dtype, X.device
        G = G.contiguous()
        gd = torch.empty(B, n, dtype=dt, device=dev)
        K.retract_vjp(X, delta, step, G, gd)
        GX = compose_left_backward(K, t.group, X, delta, step, G)
        if ctx.frozen is not None:           # frozen problems: X_new = X
            fz = ctx.frozen.bool()
            gd = gd * (~fz).to(dt).view(-1, 1)
            GX = torch.where(fz.view(1, B, *([1] * (X.dim() - 2))), G, GX)
        if ctx.dropped is not None:   # (a dropped item's step is the constant zero; its factor may be broken)
            gd = gd.masked_fill(ctx.dropped.unsqueeze(1), 0.0)
        w = ctx.solver.solve_with_snapshot(ctx.factor, gd)
        if ctx.dropped is not None:
            w = w.masked_fill(ctx.dropped.unsqueeze(1), 0.0)
        s = packed.structure
        E_, Kp = s.num_edges, s.num_priors
        new = lambda *sh: torch.zeros(*sh, dtype=dt, device=dev)  # noqa: E731
        rec, dof = tuple(X.shape[2:]), packed.dof       # group record (3,4) | (4,) | (3,3), tangent size 6 | 3 | 3
        gpi, gpj, gm, gwb = new(max(E_, 1), B, *rec), new(max(E_, 1), B, *rec), new(max(E_, 1), B, *rec), new(max(E_, 1), B, dof)
        gpp, gt, gwp = new(max(Kp, 1), B, *rec), new(max(Kp, 1), B, *rec), new(max(Kp, 1), B, dof)
        glb = new(max(E_, 1), B, 1) if t.robust_between else None
        glp = new(max(Kp, 1), B, 1) if t.robust_prior else None
        K.pg_unroll_vjp(packed.dstruct, t, w, delta, gpi, gpj, gm, gwb, gpp, gt, gwp, ell_damping=ctx.ell, g_lrb=glb, g_lrp=glp)
        # every pose's incident costs, in a fixed order: rows of [gpi ; gpj ; gpp ; 0]
        inc = packed.unroll_incidence(dev)
        src = torch.cat([gpi[:E_], gpj[:E_], gpp[:Kp], new(1, B, *rec)], 0)
        GX = GX + src[inc].sum(1)

        def fit(g, count, like):   # (count, B, ...) -> the packed input's shape (count, 1|B, ...)
            if like is None or g is None:
                return None
            g = g[:count]
            return g.sum(1, keepdim=True) if like.shape[1] == 1 and B != 1 else g
        return (None, None, None, None, GX, fit(gm, E_, t.meas), fit(gwb, E_, t.w_between), fit(gt, Kp, t.prior_target),
                fit(gwp, Kp, t.w_prior), fit(glb, E_, t.log_radius_between), fit(glp, Kp, t.log_radius_prior))
