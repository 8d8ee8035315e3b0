"""Oracle: bundle adjustment on the dense GN/LM path, restated from the reference (TEST INFRASTRUCTURE).

  Reprojection.error / .jacobians   theseus/embodied/measurements/reprojection.py:54-94
  SE3.transform_from (+ Jacobians)  theseus/geometry/se3.py:249-256 -> torchlie/functional/se3_impl.py:757-777
  Difference on SE3 / Point3        theseus/embodied/misc/local_cost_fn.py:39-61, theseus/geometry/vector.py:150-178
  RobustCostFunction, cost weights  oracle/pose_graph.py (robust_rescale, robust_weighted_error)
  DenseLinearization layout         theseus/optimizer/dense_linearization.py:29-62: columns in variable insertion
                                    order (``var_order``), rows in cost add order (``cost_order``)
State = (cameras (B,C,3,4), points (B,Np,3)); the LM loop is oracle.pose_graph.lm_optimize (problem hooks).
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from . import lie
from . import pose_graph as opg


def reprojection_jac_err(cam, X, feat, focal, k1, k2):
    """reprojection.py:54-94.  cam (...,3,4), X (...,3), feat (...,2), focal/k1/k2 (...,1) -> Jc (...,2,6), Jp (...,2,3),
    err (...,2) (unweighted)."""
    R, t = cam[..., :3], cam[..., 3]
    pc = t + (R @ X.unsqueeze(-1)).squeeze(-1)                      # se3_impl.py:757-761
    J = torch.cat([R, -R @ lie._hat(X), R], dim=-1)                 # se3_impl.py:771-774: [R, -R hat(X) | R]  (...,3,9)
    proj = -pc[..., :2] / pc[..., 2:3]
    q = (proj * proj).sum(-1, keepdim=True)
    factor = focal * (1.0 + q * (k1 + q * k2))
    dfactor = focal * (k1 + 2.0 * q * k2)
    pp = proj * factor
    d_num = J[..., 0:2, :]
    num_dden_den = pc[..., :2].unsqueeze(-1) * (J[..., 2, :] / pc[..., 2:3]).unsqueeze(-2)
    proj_jac = (num_dden_den - d_num) / pc[..., 2:].unsqueeze(-1)
    q_jac = 2.0 * proj.unsqueeze(-1) * (proj.unsqueeze(-2) @ proj_jac)
    ppj = proj_jac * factor.unsqueeze(-1) + q_jac * dfactor.unsqueeze(-1)
    return ppj[..., :6], ppj[..., 6:], pp - feat


@dataclass
class BAProblem:
    """Packed bundle-adjustment problem (examples/bundle_adjustment.py:103-160)."""

    num_cams: int
    num_points: int
    obs_cam: torch.Tensor        # (O,) long
    obs_pt: torch.Tensor         # (O,) long
    feat: torch.Tensor           # (1|B, O, 2)
    w_obs: torch.Tensor          # (1|B, O, 2) sqrt-information diagonal (ScaleCostWeight: both equal)
    focal: torch.Tensor          # (1|B, C, 1)
    k1: torch.Tensor             # (1|B, C, 1)
    k2: torch.Tensor             # (1|B, C, 1)
    cam_prior_idx: torch.Tensor  # (Kc,) long  Difference(cam, target)
    cam_prior_target: torch.Tensor  # (1|B, Kc, 3, 4)
    w_cam_prior: torch.Tensor    # (1|B, Kc, 6)
    pt_prior_idx: torch.Tensor   # (Kp,) long  Difference(point, target)
    pt_prior_target: torch.Tensor   # (1|B, Kp, 3)
    w_pt_prior: torch.Tensor     # (1|B, Kp, 3)
    var_order: List[Tuple[str, int]]    # column order: ("cam", i) | ("pt", i)  (insertion order of the objective)
    cost_order: List[Tuple[str, int]]   # row order: ("obs", o) | ("cam_prior", k) | ("pt_prior", k)
    robust_obs: Optional[str] = None
    log_radius_obs: Optional[torch.Tensor] = None   # broadcastable to (B, O, 1)
    # camera-camera Between costs (odometry; theseus/embodied/measurements/between.py:38-45): cost_order kind "cam_between"
    cc_edges: Optional[torch.Tensor] = None         # (Ecc, 2) long: Between(v0 = camera i, v1 = camera j)
    cc_meas: Optional[torch.Tensor] = None          # (1|B, Ecc, 3, 4)
    w_cc: Optional[torch.Tensor] = None             # (1|B, Ecc, 6)

    @property
    def n(self):
        return 6 * self.num_cams + 3 * self.num_points

    @property
    def m(self):
        return sum({"obs": 2, "cam_prior": 6, "pt_prior": 3, "cam_between": 6}[k] for k, _ in self.cost_order)

    def col_starts(self):
        cs, c = {}, 0
        for kind, i in self.var_order:
            cs[(kind, i)] = c
            c += 6 if kind == "cam" else 3
        return cs

    # ---- cost terms -----------------------------------------------------------------------------
    def terms(self, state):
        cams, pts = state
        Jc, Jp, e = reprojection_jac_err(cams[:, self.obs_cam], pts[:, self.obs_pt], self.feat, self.focal[:, self.obs_cam],
                                         self.k1[:, self.obs_cam], self.k2[:, self.obs_cam])
        w = self.w_obs
        Jc, Jp, e = Jc * w.unsqueeze(-1), Jp * w.unsqueeze(-1), e * w
        e_raw = e
        (Jc, Jp), e = opg.robust_rescale([Jc, Jp], e, self.robust_obs, self.log_radius_obs)
        Jcp, ecp = opg.local_jac_err(self.cam_prior_target, cams[:, self.cam_prior_idx], self.w_cam_prior)
        ept = (pts[:, self.pt_prior_idx] - self.pt_prior_target) * self.w_pt_prior     # vector.py:150-178, J = I
        return Jc, Jp, e, e_raw, Jcp, ecp, ept

    def cc_terms(self, state):
        """weighted Jacobians / error of the camera-camera Between costs: J0, J1 (B, Ecc, 6, 6), e (B, Ecc, 6), or None."""
        if self.cc_edges is None or self.cc_edges.shape[0] == 0:
            return None
        cams = state[0]
        return opg.between_jac_err(cams[:, self.cc_edges[:, 0]], cams[:, self.cc_edges[:, 1]], self.cc_meas, self.w_cc, opg.GROUPS["SE3"])

    def error_metric(self, state):
        _, _, _, e_raw, _, ecp, ept = self.terms(state)
        h = opg.robust_weighted_error(e_raw, self.robust_obs, self.log_radius_obs)
        err = 0.5 * ((h**2).sum((1, 2)) + (ecp**2).sum((1, 2)) + (ept**2).sum((1, 2)))
        cc = self.cc_terms(state)
        return err if cc is None else err + 0.5 * (cc[2]**2).sum((1, 2))

    def dense_linearize(self, state):
        cams, pts = state
        B = cams.shape[0]
        Jc, Jp, e, _, Jcp, ecp, ept = self.terms(state)
        A = torch.zeros(B, self.m, self.n, dtype=cams.dtype)
        b = torch.zeros(B, self.m, dtype=cams.dtype)
        cs = self.col_starts()
        odo = self.cc_terms(state)
        r = 0
        for kind, k in self.cost_order:
            if kind == "cam_between":
                ci, cj = cs[("cam", int(self.cc_edges[k, 0]))], cs[("cam", int(self.cc_edges[k, 1]))]
                A[:, r:r + 6, ci:ci + 6] = odo[0][:, k]
                A[:, r:r + 6, cj:cj + 6] = odo[1][:, k]
                b[:, r:r + 6] = -odo[2][:, k]
                r += 6
            elif kind == "obs":
                cc, cp = cs[("cam", int(self.obs_cam[k]))], cs[("pt", int(self.obs_pt[k]))]
                A[:, r:r + 2, cc:cc + 6] = Jc[:, k]
                A[:, r:r + 2, cp:cp + 3] = Jp[:, k]
                b[:, r:r + 2] = -e[:, k]
                r += 2
            elif kind == "cam_prior":
                cc = cs[("cam", int(self.cam_prior_idx[k]))]
                A[:, r:r + 6, cc:cc + 6] = Jcp[:, k]
                b[:, r:r + 6] = -ecp[:, k]
                r += 6
            else:
                cp = cs[("pt", int(self.pt_prior_idx[k]))]
                A[:, r:r + 3, cp:cp + 3] = torch.diag_embed(self.w_pt_prior[:, k].expand(B, 3))
                b[:, r:r + 3] = -ept[:, k]
                r += 3
        return A, b

    def retract(self, state, delta, ignore_mask=None):
        """objective.py:873-914: SE3 -> X exp(d), Point3 -> X + d (vector.py:177-178); masked problems keep X."""
        cams, pts = state
        B = cams.shape[0]
        cs = self.col_starts()
        dc = torch.stack([delta[:, cs[("cam", i)]:cs[("cam", i)] + 6] for i in range(self.num_cams)], 1)
        dp = torch.stack([delta[:, cs[("pt", i)]:cs[("pt", i)] + 3] for i in range(self.num_points)], 1)
        nc, npt = lie.se3_retract(cams, dc), pts + dp
        if ignore_mask is not None:
            nc = torch.where(ignore_mask.view(B, 1, 1, 1), cams, nc)
            npt = torch.where(ignore_mask.view(B, 1, 1), pts, npt)
        return nc, npt

    def oracle_ops(self):
        def keep_where(mask, old, new):
            B = mask.shape[0]
            return (torch.where(mask.view(B, 1, 1, 1), old[0], new[0]), torch.where(mask.view(B, 1, 1), old[1], new[1]))
        return (self.dense_linearize, self.error_metric, lambda x, d, m: self.retract(x, d, m), keep_where)
