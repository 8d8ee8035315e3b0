"""Synthetic SE3 pose-graph batches, generated on the device with the HIP SE3 ops.

Restates the data model of the reference generator
(theseus/utils/examples/pose_graph/dataset.py:238-365: random relative poses, noisy
measurements, noisy initial poses, shared DiagonalCostWeight) and of the example objective
(examples/pose_graph/pose_graph_synthetic.py:130-152: Between per edge + Difference prior on pose 0
with ScaleCostWeight(reg_w)), with a FIXED topology so that every problem in the batch shares one
structure: P poses, P-1 odometry edges (k-1,k), E-(P-1) loop closures (i,j), i<j-1 drawn once with
numpy.random.default_rng(topology_seed).
"""
from typing import Dict, List, Tuple

import numpy as np
import torch

from .. import core as th
from ..kernels import default_kernels

TRANSLATION_NOISE = 0.05
ROTATION_NOISE = 0.02
PRIOR_WEIGHT = 1e-3


def pose_graph_topology(num_poses: int, num_edges: int, topology_seed: int = 0) -> List[Tuple[int, int]]:
    P = num_poses
    edges = [(k - 1, k) for k in range(1, P)]
    if num_edges < len(edges):
        raise ValueError("num_edges must be at least num_poses - 1")
    rng = np.random.default_rng(topology_seed)
    seen = set(edges)
    while len(edges) < num_edges:
        i, j = sorted(rng.integers(0, P, size=2).tolist())
        if j - i < 2 or (i, j) in seen:
            continue
        seen.add((i, j))
        edges.append((i, j))
    return edges


def chain_graph_topology(num_poses: int, stride: int = 7, span: int = 5, seed: int = 0, shuffle: bool = True) -> List[Tuple[int, int]]:
    """A SLAM-like LARGE graph for the tile-sparse solver: the odometry chain (k, k + 1) plus a local loop closure (k, k + span)
    every ``stride`` poses -- the regime of the reference's sparse sweep (evaluations/pose_graph_synthetic.sh:5-12: up to 4096
    poses) with closures that stay local, as a trajectory's do.  ``shuffle``: pose LABELS are permuted, so the insertion order is
    far from banded and the solver's own ordering has to find the structure."""
    P = num_poses
    rng = np.random.default_rng(seed)
    edges = [(i, i + 1) for i in range(P - 1)] + [(i, i + span) for i in range(0, P - span, stride)]
    label = rng.permutation(P) if shuffle else np.arange(P)
    return [(int(label[a]), int(label[b])) for a, b in edges]


def _noise(K, B, ts, rs, dtype, device, gen):
    u = 2.0 * torch.rand(B, 6, dtype=dtype, device=device, generator=gen) - 1.0
    u[:, :3] *= ts
    u[:, 3:] *= rs
    return K.se3_exp(u)


def make_pose_graph_tensors(edges, num_poses: int, batch: int, dtype=torch.float32, device="cuda", seed: int = 1234,
                            kernels=None) -> Dict[str, torch.Tensor]:
    """name -> tensor dict (poses VERTEX_SE3__k (B,3,4), measurements EDGE_SE3__i_j (B,3,4),
    prior target, plus ground truth under GT__k)."""
    K = kernels or default_kernels()
    gen = torch.Generator(device=device).manual_seed(seed)
    B, P = batch, num_poses
    eye = torch.eye(3, 4, dtype=dtype, device=device).expand(B, 3, 4).contiguous()
    gt = [eye]
    for k in range(1, P):
        u = torch.rand(B, 6, dtype=dtype, device=device, generator=gen)
        u[:, :3] -= 0.5
        u[:, 3:] = 2.0 * u[:, 3:] - 1.0
        gt.append(K.se3_compose(gt[-1], K.se3_exp(u)))
    out = {}
    for k in range(P):
        out[f"GT__{k}"] = gt[k]
        out[f"VERTEX_SE3__{k}"] = K.se3_compose(gt[k], _noise(K, B, TRANSLATION_NOISE, ROTATION_NOISE, dtype, device, gen))
    inv = [K.se3_inverse(g) for g in gt]
    for (i, j) in edges:
        rel = K.se3_compose(inv[i], gt[j])
        out[f"EDGE_SE3__{i}_{j}"] = K.se3_compose(rel, _noise(K, B, TRANSLATION_NOISE, ROTATION_NOISE, dtype, device, gen))
    out["VERTEX_SE3__0__PRIOR"] = out["VERTEX_SE3__0"].clone()
    return out


def build_pose_graph_objective(edges, num_poses: int, dtype=torch.float32, device="cuda") -> th.Objective:
    """Objective with placeholder (batch 1) tensors; feed real ones through TheseusLayer.forward."""
    obj = th.Objective(dtype=dtype)
    eye = torch.eye(3, 4, dtype=dtype, device=device).view(1, 3, 4)
    poses = [th.SE3(tensor=eye.clone(), name=f"VERTEX_SE3__{k}") for k in range(num_poses)]
    info = torch.tensor([[1 / TRANSLATION_NOISE] * 3 + [1 / ROTATION_NOISE] * 3], dtype=dtype, device=device)
    weight = th.DiagonalCostWeight(th.Variable(info, name="EDGE_WEIGHT"))
    for (i, j) in edges:
        meas = th.SE3(tensor=eye.clone(), name=f"EDGE_SE3__{i}_{j}")
        obj.add(th.Between(poses[i], poses[j], meas, weight, name=f"between_{i}_{j}"))
    target = th.SE3(tensor=eye.clone(), name="VERTEX_SE3__0__PRIOR")
    pw = th.ScaleCostWeight(th.Variable(torch.tensor([[PRIOR_WEIGHT]], dtype=dtype, device=device), name="PRIOR_WEIGHT"))
    obj.add(th.Difference(poses[0], target, pw, name="pose_prior"))
    return obj


def input_dict(tensors: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {k: v for k, v in tensors.items() if not k.startswith("GT__")}
