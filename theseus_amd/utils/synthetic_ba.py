"""Synthetic bundle-adjustment batches (BASELINE.json configs[3]: 512 cameras / 8192 points / 32768 reprojections).

Restates the data model of the reference generator (theseus/utils/examples/bundle_adjustment/data.py:93-139,273-339:
cameras on a line looking at a band of points, per-camera observation spans with ``track_locality``, focal 1000 +- 100,
small radial terms, pixel noise) and of the example objective (examples/bundle_adjustment.py:103-160: robust Huber
Reprojection costs, Difference regularisers on every variable, strong priors on a few cameras), with ONE topology shared
by the batch; batch items = independent draws of the feature noise and of the initial perturbations.
"""
from typing import List, Tuple

import numpy as np
import torch

from .. import core as th
from ..kernels import default_kernels


def ba_topology(num_cams: int, num_points: int, track_length: int = 4, track_locality: float = 0.2,
                seed: int = 0) -> List[Tuple[int, int]]:
    """(camera, point) pairs, data.py:311-320."""
    rng = np.random.default_rng(seed)
    per_cam = track_length * num_points // num_cams
    obs = []
    for i in range(num_cams):
        span = min(per_cam + int(track_locality * num_points), num_points)
        start = (num_points - span) * i // num_cams
        for j in rng.permutation(span)[:per_cam] + start:
            obs.append((i, int(j)))
    return obs


def _project(cams, pts, focal, k1, k2):
    pc = cams[..., 3] + (cams[..., :3] @ pts.unsqueeze(-1)).squeeze(-1)
    proj = -pc[..., :2] / pc[..., 2:3]
    q = (proj * proj).sum(-1, keepdim=True)
    return proj * (focal * (1.0 + q * (k1 + q * k2)))


def make_ba_objective(num_cams=512, num_points=8192, batch=256, track_length=4, dtype=torch.float64, device="cuda", seed=0,
                      robust="huber", log_radius=1.5, reg_w=1e-4, known_cams=(0,), kernels=None):
    """-> (Objective, info dict).  Everything is generated on ``device``."""
    K = kernels or default_kernels()
    gen = torch.Generator(device=device).manual_seed(seed)
    B, C, Np = batch, num_cams, num_points
    f64 = dict(dtype=torch.float64, device=device)
    rnd = lambda *s: 2.0 * torch.rand(*s, generator=gen, **f64) - 1.0  # noqa: E731
    obs = ba_topology(C, Np, track_length, seed=seed)
    used = sorted({p for _, p in obs})
    remap = {p: k for k, p in enumerate(used)}
    # ground truth (one geometry for the batch)
    pos = torch.stack([-torch.arange(C, **f64) * 100.0 / max(C - 1, 1), torch.zeros(C, **f64), torch.full((C,), -100.0, **f64)], 1)
    tw = torch.cat([pos + rnd(C, 3), (20.0 * np.pi / 180.0) * 0.1 * rnd(C, 3)], 1)   # small rotations about identity
    gt_cams = K.se3_exp(torch.cat([torch.zeros(C, 3, **f64), tw[:, 3:]], 1).contiguous())
    gt_cams = gt_cams.clone()
    gt_cams[:, :, 3] = -(gt_cams[:, :, :3] @ tw[:, :3].unsqueeze(-1)).squeeze(-1)     # camera at position pos: t = -R p
    gt_pts = 20.0 * rnd(Np, 3) + torch.stack([torch.arange(Np, **f64) * 100.0 / Np, torch.zeros(Np, **f64), torch.zeros(Np, **f64)], 1)
    focal = 1000.0 + 100.0 * rnd(C, 1)
    k1, k2 = 0.1 * rnd(C, 1) * 1e-1, 0.05 * rnd(C, 1) * 1e-2
    oc = torch.tensor([c for c, _ in obs], device=device)
    op = torch.tensor([p for _, p in obs], device=device)
    feat = _project(gt_cams[oc], gt_pts[op], focal[oc], k1[oc], k2[oc]).unsqueeze(0) + 1.5 * rnd(B, len(obs), 2)
    pert = K.se3_exp(torch.cat([0.3 * rnd(B * C, 3), 0.01 * rnd(B * C, 3)], 1).contiguous())
    cams0 = K.se3_compose(gt_cams.unsqueeze(0).expand(B, C, 3, 4).reshape(-1, 3, 4).contiguous(), pert).view(B, C, 3, 4)
    pts0 = gt_pts.unsqueeze(0) + 0.2 * rnd(B, Np, 3)
    cast = lambda x: x.to(dtype).contiguous()  # noqa: E731
    obj = th.Objective(dtype=dtype)
    cam_v = [th.SE3(tensor=cast(cams0[:, i]), name=f"Cam{i}") for i in range(C)]
    pt_v = {p: th.Point3(tensor=cast(pts0[:, p]), name=f"Pt{p}") for p in used}
    fl = [th.Vector(tensor=cast(focal[i:i + 1]), name=f"fl{i}") for i in range(C)]
    k1v = [th.Vector(tensor=cast(k1[i:i + 1]), name=f"k1_{i}") for i in range(C)]
    k2v = [th.Vector(tensor=cast(k2[i:i + 1]), name=f"k2_{i}") for i in range(C)]
    w = th.ScaleCostWeight(torch.tensor(1.0, dtype=dtype, device=device))
    radius = th.Vector(tensor=torch.tensor([[log_radius]], dtype=dtype, device=device), name="log_loss_radius")
    loss = {"huber": th.HuberLoss, "welsch": th.WelschLoss}.get(robust)
    featc = cast(feat)
    for o, (c, p) in enumerate(obs):
        cf = th.Reprojection(cam_v[c], pt_v[p], th.Point2(tensor=featc[:, o], name=f"Feat{o}"), fl[c], k1v[c], k2v[c], weight=w,
                             name=f"reproj_{o}")
        obj.add(th.RobustCostFunction(cf, loss, radius, name=f"robust_{o}") if loss else cf)
    dw = th.ScaleCostWeight(torch.full((1,), float(np.sqrt(reg_w)), dtype=dtype, device=device))
    ident = th.SE3(tensor=torch.eye(3, 4, dtype=dtype, device=device).unsqueeze(0), name="zero_se3")
    zero = th.Point3(tensor=torch.zeros(1, 3, dtype=dtype, device=device), name="zero_point")
    for v in cam_v:
        obj.add(th.Difference(v, ident, dw, name=f"reg_{v.name}"))
    for v in pt_v.values():
        obj.add(th.Difference(v, zero, dw, name=f"reg_{v.name}"))
    cw = th.ScaleCostWeight(torch.full((1,), 100.0, dtype=dtype, device=device))
    for i in known_cams:
        obj.add(th.Difference(cam_v[i], th.SE3(tensor=cast(gt_cams[i:i + 1]), name=f"gt_cam{i}"), cw, name=f"camera_diff_{i}"))
    return obj, dict(num_cams=C, num_points=len(used), num_obs=len(obs), n=6 * C + 3 * len(used), gt_cams=gt_cams, gt_pts=gt_pts[used])
