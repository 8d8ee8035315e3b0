"""TEST-ONLY stand-in for theseus_amd.kernels.HipKernels, built on the CPU oracle.

It exists so that the host logic that does not need a GPU -- batch sharding across ranks, the LM control
flow with its batch-global predicates -- can run under ``-m "not gpu"`` (gloo, world_size 2).  The product
never imports this module: theseus_amd has no CPU path (theseus_amd/kernels.py)."""
import torch

from oracle import lie
from oracle import pose_graph as opg


LOSS = {0: None, 1: "welsch", 2: "huber"}  # theseus_amd._lib.LOSS_*


class OracleKernels:
    name = "oracle-cpu-standin"

    # ---- helpers -------------------------------------------------------------------------------
    @staticmethod
    def _problem(s, t, poses=None):
        h = s.host
        bm = lambda x: x.transpose(0, 1)  # noqa: E731  entity-major (X,B,...) -> batch-major (B,X,...)
        p = opg.PGProblem(num_poses=h.num_poses,
                          edges=torch.stack([torch.from_numpy(h.edge_i), torch.from_numpy(h.edge_j)], 1).long(),
                          meas=bm(t.meas), w_between=bm(t.w_between),
                          prior_idx=torch.from_numpy(h.prior_pose).long(),
                          prior_target=bm(t.prior_target), w_prior=bm(t.w_prior),
                          group="SE2" if t.poses.dim() == 3 else "SE3",
                          robust_between=LOSS[t.robust_between],
                          log_radius_between=bm(t.log_radius_between) if t.robust_between else None,
                          robust_prior=LOSS[t.robust_prior],
                          log_radius_prior=bm(t.log_radius_prior) if t.robust_prior else None)
        return p, bm(t.poses if poses is None else poses)

    # ---- SE3 elementwise -----------------------------------------------------------------------
    def se3_exp(self, xi, jac=False):
        assert not jac
        return lie.se3_exp(xi)

    def se3_log(self, X, jac=False):
        xi, J = lie.se3_log_jlog(X)
        return (xi, J) if jac else xi

    def se3_compose(self, X, Y):
        return lie.se3_compose(X, Y)

    def se3_inverse(self, X):
        return lie.se3_inverse(X)

    def se3_adjoint(self, X):
        return lie.se3_adjoint(X)

    # ---- pose graph ------------------------------------------------------------------------------
    def pg_assemble(self, s, t, H, g, poses=None):
        p, x = self._problem(s, t, poses)
        A, b = opg.dense_linearize(p, x)
        AtA, Atb = opg.hessian(A, b)
        n = p.n
        H[:, :n, :n] = torch.tril(AtA)
        g.copy_(Atb.squeeze(2))

    def pg_error(self, s, t, partials, err, poses=None):
        p, x = self._problem(s, t, poses)
        err.copy_(opg.error_metric(p, x))

    def pg_jacobians(self, s, t, J0, J1, eb, Jp, ep, poses=None):
        p, x = self._problem(s, t, poses)
        a, b, e, ap, e2 = opg.cost_terms(p, x)
        if p.edges.shape[0]:
            J0.copy_(a.transpose(0, 1)); J1.copy_(b.transpose(0, 1)); eb.copy_(e.transpose(0, 1))
        if p.prior_idx.shape[0]:
            Jp.copy_(ap.transpose(0, 1)); ep.copy_(e2.transpose(0, 1))

    def se3_retract(self, poses, delta, step, ignore_mask, out):
        x = poses.transpose(0, 1)
        m = ignore_mask.bool() if ignore_mask is not None else None
        out.copy_(opg.retract(x, delta * step, ignore_mask=m).transpose(0, 1))

    def retract(self, poses, delta, step, ignore_mask, out):
        if poses.dim() == 4:
            return self.se3_retract(poses, delta, step, ignore_mask, out)
        m = ignore_mask.bool() if ignore_mask is not None else None
        out.copy_(opg.retract(poses.transpose(0, 1), delta * step, ignore_mask=m).transpose(0, 1))

    # ---- SE2 elementwise -------------------------------------------------------------------------
    def se2_exp(self, xi, jac=False):
        from oracle import lie_se2
        assert not jac
        return lie_se2.se2_exp(xi)

    def se2_log(self, X, jac=False):
        from oracle import lie_se2
        xi, J = lie_se2.se2_log_jlog(X)
        return (xi, J) if jac else xi

    def se2_compose(self, X, Y):
        from oracle import lie_se2
        return lie_se2.se2_compose(X, Y)

    def se2_inverse(self, X):
        from oracle import lie_se2
        return lie_se2.se2_inverse(X)

    def se2_adjoint(self, X):
        from oracle import lie_se2
        return lie_se2.se2_adjoint(X)

    # ---- generic block assembly: dense A scatter + A^T A, as DenseLinearization does ----
    def block_assemble(self, asm, jacobians, errors, H, g):
        ref = H if H is not None else g
        B = ref.shape[0]
        m = sum(asm.cost_dims)
        A = torch.zeros(B, m, asm.n, dtype=ref.dtype)
        b = torch.zeros(B, m, dtype=ref.dtype)
        r = 0
        for c, (Js, e) in enumerate(zip(jacobians, errors)):
            d = asm.cost_dims[c]
            for s, J in enumerate(Js):
                c0, dof = asm.var_cols[asm.cost_vars[c][s]]
                A[:, r:r + d, c0:c0 + dof] = J
            b[:, r:r + d] = -e
            r += d
        if H is not None:
            H[:, :asm.n, :asm.n] = torch.tril(A.transpose(1, 2) @ A)
        if g is not None:
            g.copy_((A.transpose(1, 2) @ b.unsqueeze(2)).squeeze(2))

    # ---- implicit backward (torch autograd through the oracle, which mirrors torchlie's backward conventions) ----
    def se3_retract_vjp(self, poses, delta, step, grad_out, grad_delta):
        x = poses.transpose(0, 1)
        with torch.enable_grad():
            d = delta.detach().clone().requires_grad_(True)
            y = opg.retract(x, d * step)
            (g,) = torch.autograd.grad(y, d, grad_out.transpose(0, 1))
        grad_delta.copy_(g)

    def pg_vjp(self, s, t, w, g_meas, g_wb, g_tgt, g_wp, poses=None, g_lrb=None, g_lrp=None):
        import dataclasses
        p, x = self._problem(s, t, poses)
        B = x.shape[0]
        E, Kp = p.edges.shape[0], p.prior_idx.shape[0]
        with torch.enable_grad():
            full = lambda a: a.detach().expand(B, *a.shape[1:]).clone().requires_grad_(True)  # noqa: E731
            leaves = [full(p.meas), full(p.w_between), full(p.prior_target), full(p.w_prior)]
            lrb = full(p.log_radius_between.expand(-1, E, 1)) if p.robust_between else None
            lrp = full(p.log_radius_prior.expand(-1, Kp, 1)) if p.robust_prior else None
            pg = dataclasses.replace(p, meas=leaves[0], w_between=leaves[1], prior_target=leaves[2], w_prior=leaves[3],
                                     log_radius_between=lrb, log_radius_prior=lrp)
            leaves += [l for l in (lrb, lrp) if l is not None]
            A, b = opg.dense_linearize(pg, x)
            _, Atb = opg.hessian(A, b)
            phi = (w * Atb.squeeze(2)).sum()
            grads = torch.autograd.grad(phi, leaves, allow_unused=True)
        outs = [g_meas, g_wb, g_tgt, g_wp] + [o for o, l in ((g_lrb, lrb), (g_lrp, lrp)) if l is not None]
        for out, g, leaf in zip(outs, grads, leaves):
            if leaf.shape[1] > 0:
                out[:leaf.shape[1]].copy_((g if g is not None else torch.zeros_like(leaf)).transpose(0, 1))

    # ---- dense solver ------------------------------------------------------------------------------
    def chol_factor(self, H, n, damping, ellipsoidal, damping_eps, L, panels, info, rhs=None, y=None):
        Hl = torch.tril(H[:, :n, :n])
        A = Hl + torch.tril(Hl, -1).transpose(1, 2)
        if damping is not None:
            d = A.diagonal(dim1=1, dim2=2)
            add = damping.view(-1, 1) * d + damping_eps if ellipsoidal else damping.view(-1, 1).expand_as(d)
            A = A + torch.diag_embed(add)
        Lc, inf = torch.linalg.cholesky_ex(A)
        L[:, :n, :n] = Lc
        info.copy_(inf.to(info.dtype))
        if rhs is not None:
            y.copy_(torch.linalg.solve_triangular(Lc, rhs.unsqueeze(2), upper=False).squeeze(2))

    def chol_solve_backward(self, L, n, panels, y, x):
        x.copy_(torch.linalg.solve_triangular(L[:, :n, :n].transpose(1, 2), y.unsqueeze(2), upper=True).squeeze(2))

    def chol_solve(self, L, n, panels, rhs, x):
        x.copy_(torch.cholesky_solve(rhs.unsqueeze(2), L[:, :n, :n]).squeeze(2))

    def diag(self, H, n, d):
        d.copy_(H[:, :n, :n].diagonal(dim1=1, dim2=2))

    def lm_accept(self, delta, g, H, n, damping, prev_err, new_err, ellipsoidal, accept, down, up, reject):
        dmp = damping.view(-1, 1)
        if ellipsoidal:
            dmp = H[:, :n, :n].diagonal(dim1=1, dim2=2) * dmp
        den = (delta * (dmp * delta + g)).sum(1) / 2
        rho = (prev_err - new_err) / den
        rej = rho <= accept
        damping.copy_(torch.where(rej, damping * up, damping / down).clamp(opg.MIN_DAMPING, opg.MAX_DAMPING))
        reject.copy_(rej.to(reject.dtype))
