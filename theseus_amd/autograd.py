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
            delta = torch.empty_like(y)
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
