"""TEST-ONLY stand-in for theseus_amd.kernels.HipKernels, built on the CPU oracle.

It exists so that the host logic that does not need a GPU -- batch sharding across ranks, the LM control
flow with its batch-global predicates -- can run under ``-m "not gpu"`` (gloo, world_size 2).  The product
never imports this module: theseus_amd has no CPU path (theseus_amd/kernels.py)."""
import numpy as np
import torch

from oracle import lie
from oracle import pose_graph as opg


def loss_spec(code, table=None):
    """theseus_amd loss code(s) (_lib.LOSS_* | _lib.LOSS_FLATTEN; per-cost table) -> the oracle's loss spec."""
    one = lambda c: None if c == 0 else {1: "welsch", 2: "huber", 3: "hinge", 8: "gm"}[c & ~4] + ("+flatten" if c & 4 else "")  # noqa: E731
    return one(int(code)) if table is None else [one(int(c)) for c in table.tolist()]


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
                          group=("SO2" if t.poses.shape[-1] == 2 else "SE2") if t.poses.dim() == 3 else ("SO3" if t.poses.shape[-1] == 3 else "SE3"),
                          robust_between=loss_spec(t.robust_between, t.loss_between),
                          log_radius_between=bm(t.log_radius_between) if t.robust_between else None,
                          robust_prior=loss_spec(t.robust_prior, t.loss_prior),
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
            eb.copy_(e.transpose(0, 1))
            if J0 is not None:
                J0.copy_(a.transpose(0, 1)); J1.copy_(b.transpose(0, 1))
        if p.prior_idx.shape[0]:
            ep.copy_(e2.transpose(0, 1))
            if Jp is not None:
                Jp.copy_(ap.transpose(0, 1))

    def se3_retract(self, poses, delta, step, ignore_mask, out):
        x = poses.transpose(0, 1)
        m = ignore_mask.bool() if ignore_mask is not None else None
        out.copy_(opg.retract(x, delta[:, :6 * poses.shape[0]] * step, ignore_mask=m).transpose(0, 1))

    def retract(self, poses, delta, step, ignore_mask, out):
        if poses.dim() == 4 and poses.shape[-1] == 4:
            return self.se3_retract(poses, delta, step, ignore_mask, out)
        m = ignore_mask.bool() if ignore_mask is not None else None
        out.copy_(opg.retract(poses.transpose(0, 1), delta * step, ignore_mask=m).transpose(0, 1))

    # ---- SO3 elementwise -------------------------------------------------------------------------
    def so3_exp(self, w, jac=False):
        from oracle import lie_so3
        R, J = lie_so3.so3_exp_jexp(w)
        return (R, J) if jac else R

    def so3_log(self, R, jac=False):
        from oracle import lie_so3
        w, J = lie_so3.so3_log_jlog(R)
        return (w, J) if jac else w

    def so3_compose(self, X, Y):
        return X @ Y

    def so3_inverse(self, X):
        return X.transpose(-1, -2).contiguous()

    def so3_adjoint(self, X):
        return X.clone()

    # ---- SO2 elementwise -------------------------------------------------------------------------
    def so2_exp(self, theta, jac=False):
        from oracle import lie_so2
        X, J = lie_so2.so2_exp_jexp(theta)
        return (X, J) if jac else X

    def so2_log(self, X, jac=False):
        from oracle import lie_so2
        th_, J = lie_so2.so2_log_jlog(X)
        return (th_, J) if jac else th_

    def so2_compose(self, X, Y):
        from oracle import lie_so2
        return lie_so2.so2_compose(X, Y)

    def so2_inverse(self, X):
        from oracle import lie_so2
        return lie_so2.so2_inverse(X)

    def so2_adjoint(self, X):
        from oracle import lie_so2
        return lie_so2.so2_adjoint(X)

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

    # ---- bundle adjustment: block quantities from the oracle's per-cost terms (oracle/ba.py) ------------------
    @staticmethod
    def _ba_problem(s, t, cams=None, points=None):
        from oracle import ba as oba
        h = s.host
        bm = lambda x: x.transpose(0, 1)  # noqa: E731
        lg = lambda a: torch.from_numpy(h.t[a].astype("int64"))  # noqa: E731
        p = oba.BAProblem(
            num_cams=h.num_cams, num_points=h.num_points, obs_cam=lg("obs_cam")[:h.num_obs], obs_pt=lg("obs_pt")[:h.num_obs],
            feat=bm(t.feat), w_obs=bm(t.w_obs), focal=bm(t.focal), k1=bm(t.k1), k2=bm(t.k2),
            cam_prior_idx=lg("cam_prior_cam")[:h.num_cam_priors], cam_prior_target=bm(t.cam_prior_target),
            w_cam_prior=bm(t.w_cam_prior), pt_prior_idx=lg("pt_prior_pt")[:h.num_pt_priors],
            pt_prior_target=bm(t.pt_prior_target), w_pt_prior=bm(t.w_pt_prior), var_order=[], cost_order=[],
            robust_obs=loss_spec(t.robust_obs), log_radius_obs=bm(t.log_radius_obs) if t.robust_obs else None)
        return p, (bm(t.cams if cams is None else cams), bm(t.points if points is None else points))

    def ba_assemble(self, s, t, Hcc, Hpp, W, gd, g, diag):
        p, state = self._ba_problem(s, t)
        Jc, Jp, e, _, Jcp, ecp, ept = p.terms(state)
        B, C, Np = state[0].shape[0], p.num_cams, p.num_points
        hcc = torch.zeros(B, C, 6, 6, dtype=g.dtype).index_add_(1, p.obs_cam, Jc.transpose(2, 3) @ Jc)
        hcc.index_add_(1, p.cam_prior_idx, (Jcp.transpose(2, 3) @ Jcp).expand(B, -1, 6, 6))
        hpp = torch.zeros(B, Np, 3, 3, dtype=g.dtype).index_add_(1, p.obs_pt, Jp.transpose(2, 3) @ Jp)
        hpp.index_add_(1, p.pt_prior_idx, torch.diag_embed(p.w_pt_prior ** 2).expand(B, -1, 3, 3))
        gc = torch.zeros(B, C, 6, dtype=g.dtype).index_add_(1, p.obs_cam, -(Jc.transpose(2, 3) @ e.unsqueeze(3)).squeeze(3))
        gc.index_add_(1, p.cam_prior_idx, -(Jcp.transpose(2, 3) @ ecp.unsqueeze(3)).squeeze(3).expand(B, -1, 6))
        gp = torch.zeros(B, Np, 3, dtype=g.dtype).index_add_(1, p.obs_pt, -(Jp.transpose(2, 3) @ e.unsqueeze(3)).squeeze(3))
        gp.index_add_(1, p.pt_prior_idx, -(p.w_pt_prior * ept).expand(B, -1, 3))
        planar = lambda x: x.reshape(x.shape[0], x.shape[1], -1).permute(1, 2, 0)  # noqa: E731  (B, N, ...) -> (N, comps, B)
        Hcc.copy_(planar(hcc))
        iu = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
        Hpp.copy_(planar(torch.stack([hpp[:, :, i, j] for i, j in iu], -1)))
        if p.obs_cam.numel():
            W[:p.obs_cam.numel()].copy_(planar(Jc.transpose(2, 3) @ Jp))
        g[:, :6 * C] = gc.reshape(B, -1)
        g[:, 6 * C:] = gp.reshape(B, -1)
        gd.copy_(g)
        diag[:, :6 * C] = hcc.diagonal(dim1=2, dim2=3).reshape(B, -1)
        diag[:, 6 * C:] = hpp.diagonal(dim1=2, dim2=3).reshape(B, -1)

    def ba_av(self, s, t, v, rows, out_t):
        p, state = self._ba_problem(s, t)
        Jc, Jp, _, _, Jcp, _, _ = p.terms(state)
        B, C, Np = state[0].shape[0], p.num_cams, p.num_points
        vc, vp = v[:, :6 * C].reshape(B, C, 6), v[:, 6 * C:].reshape(B, Np, 3)
        out = torch.zeros(B, out_t.shape[0], dtype=v.dtype)

        def put(first_rows, vals):   # vals (B, K, d) -> rows first_rows[k] + [0, d)
            K, d = vals.shape[1], vals.shape[2]
            idx = (first_rows[:K].long().view(-1, 1) + torch.arange(d)).view(-1)
            out[:, idx] = vals.reshape(B, -1)
        if p.obs_cam.numel():
            put(rows[0], (Jc @ vc[:, p.obs_cam].unsqueeze(3) + Jp @ vp[:, p.obs_pt].unsqueeze(3)).squeeze(3))
        if p.cam_prior_idx.numel():
            put(rows[1], (Jcp @ vc[:, p.cam_prior_idx].unsqueeze(3)).squeeze(3))
        if p.pt_prior_idx.numel():
            put(rows[2], p.w_pt_prior * vp[:, p.pt_prior_idx])
        out_t.copy_(out.t())

    def ba_vjp(self, s, t, w, grads):
        """d(w^T g)/d theta by torch autograd through the oracle's restatement of the cost terms."""
        import dataclasses
        p, state = self._ba_problem(s, t)
        B, C, Np, O = state[0].shape[0], p.num_cams, p.num_points, p.obs_cam.numel()
        with torch.enable_grad():
            full = lambda a: a.detach().expand(B, *a.shape[1:]).clone().requires_grad_(True)  # noqa: E731
            lv = dict(feat=full(p.feat), w_obs=full(p.w_obs), focal=full(p.focal), k1=full(p.k1), k2=full(p.k2),
                      cam_prior_target=full(p.cam_prior_target), w_cam_prior=full(p.w_cam_prior),
                      pt_prior_target=full(p.pt_prior_target), w_pt_prior=full(p.w_pt_prior))
            if p.robust_obs:
                lv["log_radius"] = full(p.log_radius_obs.expand(-1, O, 1))
            pg = dataclasses.replace(p, feat=lv["feat"], w_obs=lv["w_obs"], focal=lv["focal"], k1=lv["k1"], k2=lv["k2"],
                                     cam_prior_target=lv["cam_prior_target"], w_cam_prior=lv["w_cam_prior"],
                                     pt_prior_target=lv["pt_prior_target"], w_pt_prior=lv["w_pt_prior"],
                                     log_radius_obs=lv.get("log_radius"))
            Jc, Jp, e, _, Jcp, ecp, ept = pg.terms(state)
            gc = torch.zeros(B, C, 6, dtype=w.dtype).index_add(1, p.obs_cam, -(Jc.transpose(2, 3) @ e.unsqueeze(3)).squeeze(3))
            gc = gc.index_add(1, p.cam_prior_idx, -(Jcp.transpose(2, 3) @ ecp.unsqueeze(3)).squeeze(3).expand(B, -1, 6))
            gp = torch.zeros(B, Np, 3, dtype=w.dtype).index_add(1, p.obs_pt, -(Jp.transpose(2, 3) @ e.unsqueeze(3)).squeeze(3))
            gp = gp.index_add(1, p.pt_prior_idx, -(pg.w_pt_prior * ept).expand(B, -1, 3))
            phi = (w[:, :6 * C] * gc.reshape(B, -1)).sum() + (w[:, 6 * C:] * gp.reshape(B, -1)).sum()
            names = list(lv)
            gr = dict(zip(names, torch.autograd.grad(phi, [lv[k] for k in names], allow_unused=True)))
        for k, out in grads.items():
            if out is None or k not in gr:
                continue
            g = gr[k] if gr[k] is not None else torch.zeros_like(lv[k])
            if k in ("focal", "k1", "k2"):
                # the kernel reports calibration gradients PER OBSERVATION; any split that sums to the camera's works
                first = {}
                for o, c in enumerate(p.obs_cam.tolist()):
                    first.setdefault(c, o)
                out.zero_()
                for c, o in first.items():
                    out[o] = g[:, c, 0]
            elif g.shape[1] > 0:
                out[:g.shape[1]].copy_(g.transpose(0, 1))

    def ba_unroll_vjp(self, s, t, w, delta, grads, ell_damping=None):
        """thx_ba_unroll_vjp: per cost, the gradient of Phi = -(J w) . (r + J delta) [- lambda sum_k w_k delta_k H_kk] (J, r with the
        cost weight and the robust rescale) by torch autograd through the oracle's Reprojection / Difference formulas; every cost
        owns a copy of its camera / point, so the gradients come out per cost as the kernel reports them."""
        import dataclasses
        from oracle import ba as oba
        p, (cams, pts) = self._ba_problem(s, t)
        B, C, Np, O = cams.shape[0], p.num_cams, p.num_points, p.obs_cam.numel()
        Kc, Kp = p.cam_prior_idx.numel(), p.pt_prior_idx.numel()
        wc, wp = w[:, :6 * C].reshape(B, C, 6), w[:, 6 * C:].reshape(B, Np, 3)
        dc, dp = delta[:, :6 * C].reshape(B, C, 6), delta[:, 6 * C:].reshape(B, Np, 3)
        lam = None if ell_damping is None else ell_damping.view(B, 1, 1)
        mv = lambda J, v: (J @ v.unsqueeze(-1)).squeeze(-1)   # noqa: E731
        with torch.enable_grad():
            full = lambda a: a.detach().expand(B, *a.shape[1:]).clone().requires_grad_(True)  # noqa: E731
            lv = dict(cam_obs=full(cams[:, p.obs_cam]), pt_obs=full(pts[:, p.obs_pt]), feat=full(p.feat), w_obs=full(p.w_obs),
                      focal=full(p.focal[:, p.obs_cam]), k1=full(p.k1[:, p.obs_cam]), k2=full(p.k2[:, p.obs_cam]),
                      cam_prior_cam=full(cams[:, p.cam_prior_idx]), cam_prior_target=full(p.cam_prior_target),
                      w_cam_prior=full(p.w_cam_prior), pt_prior_pt=full(pts[:, p.pt_prior_idx]),
                      pt_prior_target=full(p.pt_prior_target), w_pt_prior=full(p.w_pt_prior))
            if p.robust_obs:
                lv["log_radius_obs"] = full(p.log_radius_obs.expand(-1, O, 1))
            phi = cams.new_zeros(())
            if O:
                Jc, Jp, e = oba.reprojection_jac_err(lv["cam_obs"], lv["pt_obs"], lv["feat"], lv["focal"], lv["k1"], lv["k2"])
                ws = lv["w_obs"]
                Jc, Jp, e = Jc * ws.unsqueeze(-1), Jp * ws.unsqueeze(-1), e * ws
                (Jc, Jp), e = opg.robust_rescale([Jc, Jp], e, p.robust_obs, lv.get("log_radius_obs"))
                wco, wpo, dco, dpo = wc[:, p.obs_cam], wp[:, p.obs_pt], dc[:, p.obs_cam], dp[:, p.obs_pt]
                phi = phi - ((mv(Jc, wco) + mv(Jp, wpo)) * (e + mv(Jc, dco) + mv(Jp, dpo))).sum()
                if lam is not None:
                    phi = phi - (lam * ((Jc ** 2).sum(-2) * wco * dco)).sum() - (lam * ((Jp ** 2).sum(-2) * wpo * dpo)).sum()
            if Kc:
                Jq, eq = opg.local_jac_err(lv["cam_prior_target"], lv["cam_prior_cam"], lv["w_cam_prior"])
                wq, dq = wc[:, p.cam_prior_idx], dc[:, p.cam_prior_idx]
                phi = phi - (mv(Jq, wq) * (eq + mv(Jq, dq))).sum()
                if lam is not None:
                    phi = phi - (lam * (Jq ** 2).sum(-2) * wq * dq).sum()
            if Kp:
                sw = lv["w_pt_prior"]
                et = (lv["pt_prior_pt"] - lv["pt_prior_target"]) * sw
                wq, dq = wp[:, p.pt_prior_idx], dp[:, p.pt_prior_idx]
                phi = phi - (sw * wq * (et + sw * dq)).sum()
                if lam is not None:
                    phi = phi - (lam * sw ** 2 * wq * dq).sum()
            names = list(lv)
            gr = dict(zip(names, torch.autograd.grad(phi, [lv[k] for k in names], allow_unused=True)))
        for k, out in grads.items():
            if out is None or k not in gr or lv[k].shape[1] == 0:
                continue
            g = gr[k] if gr[k] is not None else torch.zeros_like(lv[k])
            if k in ("focal", "k1", "k2"):
                g = g.squeeze(-1)
            out[:g.shape[1]].copy_(g.transpose(0, 1))

    @staticmethod
    def _sym3(h):  # (..., 6) -> (..., 3, 3)
        return torch.stack([torch.stack([h[..., 0], h[..., 1], h[..., 2]], -1), torch.stack([h[..., 1], h[..., 3], h[..., 4]], -1),
                            torch.stack([h[..., 2], h[..., 4], h[..., 5]], -1)], -2)

    def ba_schur(self, s, Hcc, Hpp, W, g, damping, ellipsoidal, damping_eps, S, rhs, Hinv, tvec, info):
        h = s.host
        C, Np, O = h.num_cams, h.num_points, h.num_obs
        B = g.shape[0]
        oc, op = (torch.from_numpy(h.t[k].astype("int64"))[:O] for k in ("obs_cam", "obs_pt"))
        batch_major = lambda x, *sh: x.permute(2, 0, 1).reshape(x.shape[2], x.shape[0], *sh)  # noqa: E731  planar -> (B, N, ...)
        hcc, hpp = batch_major(Hcc, 6, 6).clone(), self._sym3(batch_major(Hpp, 6))
        if damping is not None:
            lam = damping.view(B, 1, 1)
            dc, dp = hcc.diagonal(dim1=2, dim2=3), hpp.diagonal(dim1=2, dim2=3)
            hcc = hcc + torch.diag_embed(lam * dc + damping_eps if ellipsoidal else lam.expand_as(dc))
            hpp = hpp + torch.diag_embed(lam * dp + damping_eps if ellipsoidal else lam.expand_as(dp))
        _, inf = torch.linalg.cholesky_ex(hpp)
        info.copy_((inf != 0).any(1).to(info.dtype))
        hi = torch.linalg.inv(hpp)
        Hinv.copy_(torch.stack([hi[..., 0, 0], hi[..., 0, 1], hi[..., 0, 2], hi[..., 1, 1], hi[..., 1, 2], hi[..., 2, 2]], -1).permute(1, 2, 0))
        gp = g[:, 6 * C:].reshape(B, Np, 3)
        tv = (hi @ gp.unsqueeze(3)).squeeze(3)
        tvec.copy_(tv.permute(1, 2, 0))
        Wb = batch_major(W[:O], 6, 3)                                     # (B,O,6,3)
        Hcp = torch.zeros(B, 6 * C, 3 * Np, dtype=g.dtype)
        for o in range(O):
            c, p_ = int(oc[o]), int(op[o])
            Hcp[:, 6 * c:6 * c + 6, 3 * p_:3 * p_ + 3] += Wb[:, o]
        HccD = torch.zeros(B, 6 * C, 6 * C, dtype=g.dtype)
        HpiD = torch.zeros(B, 3 * Np, 3 * Np, dtype=g.dtype)
        for c in range(C):
            HccD[:, 6 * c:6 * c + 6, 6 * c:6 * c + 6] = hcc[:, c]
        for p_ in range(Np):
            HpiD[:, 3 * p_:3 * p_ + 3, 3 * p_:3 * p_ + 3] = hi[:, p_]
        Sfull = HccD - Hcp @ HpiD @ Hcp.transpose(1, 2)
        S[:, :6 * C, :6 * C] = torch.tril(Sfull)
        rhs.copy_(g[:, :6 * C] - (Hcp @ tv.reshape(B, -1, 1)).squeeze(2))

    def ba_backsub(self, s, W, Hinv, tvec, delta):
        h = s.host
        C, Np, O = h.num_cams, h.num_points, h.num_obs
        B = delta.shape[0]
        oc, op = (torch.from_numpy(h.t[k].astype("int64"))[:O] for k in ("obs_cam", "obs_pt"))
        dc = delta[:, :6 * C].reshape(B, C, 6)
        batch_major = lambda x, *sh: x.permute(2, 0, 1).reshape(x.shape[2], x.shape[0], *sh)  # noqa: E731  planar -> (B, N, ...)
        acc = torch.zeros(B, Np, 3, dtype=delta.dtype).index_add_(
            1, op, (batch_major(W[:O], 6, 3).transpose(2, 3) @ dc[:, oc].unsqueeze(3)).squeeze(3))
        hi = self._sym3(batch_major(Hinv, 6))
        delta[:, 6 * C:] = (batch_major(tvec, 3) - (hi @ acc.unsqueeze(3)).squeeze(3)).reshape(B, -1)

    def ba_error(self, s, t, partials, err, cams=None, points=None):
        p, state = self._ba_problem(s, t, cams, points)
        err.copy_(p.error_metric(state))

    def copy_where(self, mask, src, dst):
        m = mask.bool().view(1, -1, *([1] * (dst.dim() - 2)))
        dst.copy_(torch.where(m, src, dst))

    def vec_retract(self, x, delta, col0, step, ignore_mask, out):
        N, B, dof = x.shape
        new = x + (delta[:, col0:col0 + N * dof] * step).reshape(B, N, dof).transpose(0, 1)
        if ignore_mask is not None:
            new = torch.where(ignore_mask.bool().view(1, B, 1), x, new)
        out.copy_(new)

    def lm_accept_diag(self, delta, g, diag, n, damping, prev_err, new_err, ellipsoidal, accept, down, up, reject):
        dmp = damping.view(-1, 1)
        if ellipsoidal:
            dmp = diag[:, :n] * dmp
        den = (delta * (dmp * delta + g)).sum(1) / 2
        rej = (prev_err - new_err) / den <= accept
        damping.copy_(torch.where(rej, damping * up, damping / down).clamp(1e-7, 1e7))
        reject.copy_(rej.to(reject.dtype))

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
        ncol = (6 if poses.dim() == 4 and poses.shape[-1] == 4 else 3) * poses.shape[0]   # (bundle adjustment: camera columns)
        with torch.enable_grad():
            d = delta[:, :ncol].detach().clone().requires_grad_(True)
            y = opg.retract(x, d * step)
            (g,) = torch.autograd.grad(y, d, grad_out.transpose(0, 1))
        grad_delta[:, :ncol].copy_(g)

    def retract_vjp(self, poses, delta, step, grad_out, grad_delta):
        return self.se3_retract_vjp(poses, delta, step, grad_out, grad_delta)   # opg.retract dispatches on the shape

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

    def pg_unroll_vjp(self, s, t, w, delta, g_pose_i, g_pose_j, g_meas, g_wb, g_pose_p, g_tgt, g_wp, poses=None, ell_damping=None,
                      g_lrb=None, g_lrp=None):
        """thx_pg_unroll_vjp: per cost, the gradient of Phi = -(J w) . (r + J delta) [- lambda sum_i w_i delta_i H_ii] (J, r with the
        cost weight and the robust rescale) by torch autograd through the oracle's Between / Local formulas (which carry the
        reference's autograd conventions)."""
        p, x = self._problem(s, t, poses)
        B = x.shape[0]
        E, Kp = p.edges.shape[0], p.prior_idx.shape[0]
        i, j = p.edges[:, 0], p.edges[:, 1]
        blk = lambda v, idx: v.view(B, -1, p.dof)[:, idx]   # noqa: E731   (B, n) -> (B, len(idx), dof)
        with torch.enable_grad():
            full = lambda a: a.detach().expand(B, *a.shape[1:]).clone().requires_grad_(True)  # noqa: E731
            v0, v1, vp = full(x[:, i]), full(x[:, j]), full(x[:, p.prior_idx])
            meas, wb, tgt, wp = full(p.meas), full(p.w_between), full(p.prior_target), full(p.w_prior)
            lrb = full(p.log_radius_between.expand(-1, E, 1)) if p.robust_between else None
            lrp = full(p.log_radius_prior.expand(-1, Kp, 1)) if p.robust_prior else None
            mv = lambda J, v: (J @ v.unsqueeze(-1)).squeeze(-1)   # noqa: E731
            lam = None if ell_damping is None else ell_damping.view(B, 1, 1)
            phi = x.new_zeros(())
            if E:
                J0, J1, eb = opg.between_jac_err(v0, v1, meas, wb, p.G)
                (J0, J1), eb = opg.robust_rescale([J0, J1], eb, p.robust_between, lrb)
                phi = phi - ((mv(J0, blk(w, i)) + mv(J1, blk(w, j))) * (eb + mv(J0, blk(delta, i)) + mv(J1, blk(delta, j)))).sum()
                if lam is not None:   # ellipsoidal damping: -lambda sum_i w_i delta_i H_ii
                    phi = phi - (lam * ((J0 ** 2).sum(-2) * blk(w, i) * blk(delta, i) + (J1 ** 2).sum(-2) * blk(w, j) * blk(delta, j))).sum()
            if Kp:
                Jp, ep = opg.local_jac_err(tgt, vp, wp, p.G)
                (Jp,), ep = opg.robust_rescale([Jp], ep, p.robust_prior, lrp)
                phi = phi - (mv(Jp, blk(w, p.prior_idx)) * (ep + mv(Jp, blk(delta, p.prior_idx)))).sum()
                if lam is not None:
                    phi = phi - (lam * (Jp ** 2).sum(-2) * blk(w, p.prior_idx) * blk(delta, p.prior_idx)).sum()
            leaves = [v0, v1, meas, wb, vp, tgt, wp] + [l for l in (lrb, lrp) if l is not None]
            grads = torch.autograd.grad(phi, leaves, allow_unused=True)
        outs = [g_pose_i, g_pose_j, g_meas, g_wb, g_pose_p, g_tgt, g_wp] + [o for o, l in ((g_lrb, lrb), (g_lrp, lrp)) if l is not None]
        for out, g, leaf in zip(outs, grads, leaves):
            if out is not None and leaf.shape[1] > 0:
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

    def chol_factor_sparse(self, H, n, damping, ellipsoidal, damping_eps, L, panels, info, pattern, rhs=None, y=None):
        """Dense factorisation + the check that makes the CPU run meaningful: the numeric factor has no entry outside the
        symbolic tile pattern the HIP kernels would have been launched over."""
        self.chol_factor(H, n, damping, ellipsoidal, damping_eps, L, panels, info, rhs=rhs, y=y)
        nt = pattern.ntiles
        Lp = torch.nn.functional.pad(L[:, :n, :n], (0, nt * 128 - n, 0, nt * 128 - n))
        tiles = Lp.view(L.shape[0], nt, 128, nt, 128).abs().amax(dim=(0, 2, 4)) > 0
        assert not (tiles & ~torch.from_numpy(pattern.lower)).any(), "numeric fill outside the symbolic tile pattern"

    def chol_solve_sparse(self, L, n, panels, rhs, x, pattern, backward_only=False):
        """The list-driven solves read ONLY the tiles of the row lists: emulate that by masking L to the pattern (a factor with
        fill outside it would give a different answer here, as it would on the GPU)."""
        nt = pattern.ntiles
        rows = np.zeros((nt, nt), dtype=bool)
        rp, rt = pattern.tables["row_ptr"], pattern.tables["row_tile"]
        for i in range(nt):
            rows[i, rt[rp[i]:rp[i + 1]]] = True
            rows[i, i] = True
        assert (rows == pattern.lower).all(), "row lists disagree with the column tables"
        mask = torch.from_numpy(np.kron(rows, np.ones((128, 128), dtype=bool))[:n, :n])
        Lm = torch.where(mask, L[:, :n, :n], torch.zeros((), dtype=L.dtype))
        if backward_only:
            x.copy_(torch.linalg.solve_triangular(Lm.transpose(1, 2), rhs.unsqueeze(2), upper=True).squeeze(2))
        else:
            x.copy_(torch.cholesky_solve(rhs.unsqueeze(2), Lm).squeeze(2))

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
