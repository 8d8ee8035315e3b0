"""g2o pose-graph files and dataset batching: the data format on the input side of the hot path.

Mirrors theseus/utils/examples/pose_graph/dataset.py: ``PoseGraphEdge`` (:14-30), ``read_3D_g2o_file`` (:35-104),
``read_2D_g2o_file`` (:110-172), ``PoseGraphDataset`` (:175-235, 367-439) incl. ``write_3D_g2o`` and
``get_batch_dataset`` -- same names, same argument meaning, same variable names (``VERTEX_SE3__i``, ``EDGE_SE3__n``,
``EDGE_WEIGHT__n``), so the objectives of examples/pose_graph/pose_graph_{benchmark,cube}.py build unchanged on top.

File conventions (g2o wiki, SLAM-3D / SLAM-2D):
  VERTEX_SE3:QUAT i  x y z  qx qy qz qw
  EDGE_SE3:QUAT  i j  x y z  qx qy qz qw  <21 upper-triangular entries of the 6x6 information matrix, row major>
  VERTEX_SE2 i  x y theta
  EDGE_SE2  i j  x y theta  <6 upper-triangular entries of the 3x3 information matrix>
Quaternions are normalised on read and reordered to (w, x, y, z), the order of ``SE3(x_y_z_quaternion=...)``; the cost
weight is the SQUARE ROOT of the information DIAGONAL (entries 0, 6, 11, 15, 18, 20 of the 21; 0, 3, 5 of the 6) -- the
off-diagonal information is ignored, as in the reference.

Reference quirk: its 2-D reader builds the weight with ``np.array(1, tokens[6:], ...)`` (dataset.py:137), which raises a
TypeError on every EDGE_SE2 line, so 2-D files with edges cannot be read there at all; this reader implements what the
3-D twin does (sqrt of the information diagonal).
"""
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .. import core as th

Pose = Union[th.SE2, th.SE3]

_INFO_DIAG_3D = [0, 6, 11, 15, 18, 20]
_INFO_DIAG_2D = [0, 3, 5]


class PoseGraphEdge:
    # dataset.py:14-30
    def __init__(self, i: int, j: int, relative_pose: Pose, weight: Optional[th.DiagonalCostWeight] = None):
        self.i = i
        self.j = j
        self.relative_pose = relative_pose
        self.weight = weight

    def to(self, *args, **kwargs):
        self.weight.to(*args, **kwargs)
        self.relative_pose.to(*args, **kwargs)


def _floats(tokens: Sequence[str], dtype) -> torch.Tensor:
    return torch.from_numpy(np.array([tokens], dtype=np.float64)).to(dtype)


def _quat_wxyz(x_y_z_q: torch.Tensor) -> torch.Tensor:
    """(1,7) [x y z qx qy qz qw] -> [x y z qw qx qy qz] with a unit quaternion (dataset.py:55-56, 83-84)."""
    out = x_y_z_q.clone()
    out[:, 3:] /= torch.norm(out[:, 3:], dim=1)
    out[:, 3:] = out[:, [6, 3, 4, 5]]
    return out


def _sqrt_information(tokens: Sequence[str], sel: List[int], dtype) -> th.Variable:
    info = np.array(tokens, dtype=np.float64)
    if info.shape[0] <= sel[-1]:
        raise ValueError(f"g2o edge: expected {sel[-1] + 1} information entries, found {info.shape[0]}")
    return th.Variable(torch.from_numpy(info[sel]).to(dtype).sqrt().view(1, -1))


def _read(path: str, dtype, vertex_tag: str, edge_tag: str, n_pose: int, make_pose, sel, prefix: str):
    num_vertices = 0
    verts = dict()
    edges: List[PoseGraphEdge] = []
    with open(path, "r") as file:
        for line in file:
            tokens = line.split()
            if not tokens:
                continue
            if tokens[0] == edge_tag:
                i, j = int(tokens[1]), int(tokens[2])
                n = len(edges)
                relative_pose = make_pose(_floats(tokens[3:3 + n_pose], dtype), f"EDGE_{prefix}__{n}")
                weight = th.DiagonalCostWeight(_sqrt_information(tokens[3 + n_pose:], sel, dtype), name=f"EDGE_WEIGHT__{n}")
                edges.append(PoseGraphEdge(i, j, relative_pose, weight))
                num_vertices = max(num_vertices, i, j)
            elif tokens[0] == vertex_tag:
                i = int(tokens[1])
                verts[i] = _floats(tokens[2:2 + n_pose], dtype)
                num_vertices = max(num_vertices, i)
    num_vertices += 1
    vertices = [make_pose(v, f"VERTEX_{prefix}__{i}") for i, v in sorted(verts.items())]
    return num_vertices, vertices, edges


def read_3D_g2o_file(path: str, dtype: Optional[torch.dtype] = None) -> Tuple[int, List[th.SE3], List[PoseGraphEdge]]:
    """(number of poses, initial values, edges) of a SLAM-3D g2o file (dataset.py:35-104)."""
    return _read(path, dtype, "VERTEX_SE3:QUAT", "EDGE_SE3:QUAT", 7,
                 lambda v, name: th.SE3(x_y_z_quaternion=_quat_wxyz(v), name=name), _INFO_DIAG_3D, "SE3")


def read_2D_g2o_file(path: str, dtype: Optional[torch.dtype] = None) -> Tuple[int, List[th.SE2], List[PoseGraphEdge]]:
    """(number of poses, initial values, edges) of a SLAM-2D g2o file (dataset.py:110-172; see the module docstring for
    the reference's weight bug)."""
    return _read(path, dtype, "VERTEX_SE2", "EDGE_SE2", 3, lambda v, name: th.SE2(x_y_theta=v, name=name),
                 _INFO_DIAG_2D, "SE2")


def rotation_to_quaternion(R: torch.Tensor) -> torch.Tensor:
    """(N,3,3) rotation matrices -> (N,4) unit quaternions (w, x, y, z), w >= 0.  Host-side I/O helper (not on the hot
    path): the numerically safe branch per matrix is picked by the largest of (trace, R00, R11, R22)."""
    R = R.double()
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    cand = torch.stack([
        torch.stack([1 + m00 + m11 + m22, R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], 1),
        torch.stack([R[:, 2, 1] - R[:, 1, 2], 1 + m00 - m11 - m22, R[:, 0, 1] + R[:, 1, 0], R[:, 0, 2] + R[:, 2, 0]], 1),
        torch.stack([R[:, 0, 2] - R[:, 2, 0], R[:, 0, 1] + R[:, 1, 0], 1 - m00 + m11 - m22, R[:, 1, 2] + R[:, 2, 1]], 1),
        torch.stack([R[:, 1, 0] - R[:, 0, 1], R[:, 0, 2] + R[:, 2, 0], R[:, 1, 2] + R[:, 2, 1], 1 - m00 - m11 + m22], 1),
    ], 1)  # (N, 4 candidates, 4): candidate k is 4 q_k * q
    k = torch.stack([m00 + m11 + m22, m00, m11, m22], 1).argmax(1)
    q = cand[torch.arange(R.shape[0]), k]
    q = q / q.norm(dim=1, keepdim=True)
    return torch.where(q[:, :1] < 0, -q, q)


def _fmt(values) -> str:
    return " ".join(repr(float(x)) for x in values)  # shortest round-trip decimal of the fp64 value


class PoseGraphDataset:
    """A batch of pose graphs sharing one topology (dataset.py:175-235, 367-439): ``poses[k].tensor`` and
    ``edges[e].relative_pose.tensor`` are (dataset_size, ...) tensors; ``get_batch_dataset(b)`` slices batch ``b``."""

    def __init__(self, poses: List[Pose], edges: List[PoseGraphEdge], gt_poses: Optional[List[Pose]] = None,
                 batch_size: int = 1, device=None):
        sizes = [p.shape[0] for p in poses]
        if gt_poses is not None:
            sizes.extend(p.shape[0] for p in gt_poses)
        sizes.extend(e.relative_pose.shape[0] for e in edges)
        if len(set(sizes)) != 1:
            raise ValueError(f"poses, ground-truth poses and edge measurements must share one dataset size, got {sorted(set(sizes))}")
        self.poses = poses
        self.edges = edges
        self.gt_poses = gt_poses
        self.batch_size = batch_size
        self.dataset_size = sizes[0]
        self.num_batches = (self.dataset_size - 1) // self.batch_size + 1
        self.to(device=device)

    @staticmethod
    def load_3D_g2o_file(path: str, dtype: Optional[torch.dtype] = None) -> "PoseGraphDataset":
        _, poses, edges = read_3D_g2o_file(path, dtype)
        return PoseGraphDataset(poses, edges)

    @staticmethod
    def load_2D_g2o_file(path: str, dtype: Optional[torch.dtype] = None) -> "PoseGraphDataset":
        _, poses, edges = read_2D_g2o_file(path, dtype)
        return PoseGraphDataset(poses, edges)

    def write_3D_g2o(self, filename: str):
        """One file ``<filename>_<n>.g2o`` per dataset item (dataset.py:367-399): edges first, then vertices; the
        information diagonal is the squared cost weight."""
        for n in range(self.dataset_size):
            with open(filename + f"_{n}.g2o", "w") as file:
                for edge in self.edges:
                    m = edge.relative_pose.tensor[n:n + 1].detach().cpu()
                    q = rotation_to_quaternion(m[:, :, :3])[0].numpy()
                    t = m[0, :, 3].double().numpy()
                    w = (edge.weight.diagonal.tensor.detach().cpu().double() ** 2)[0].numpy()
                    info = np.zeros(21)
                    info[_INFO_DIAG_3D] = w
                    file.write(f"EDGE_SE3:QUAT {edge.i} {edge.j} " + _fmt(t) + " " + _fmt(q[[1, 2, 3, 0]]) + " " +
                               _fmt(info) + "\n")
                for i, pose in enumerate(self.poses):
                    p = pose.tensor[n:n + 1].detach().cpu()
                    q = rotation_to_quaternion(p[:, :, :3])[0].numpy()
                    t = p[0, :, 3].double().numpy()
                    file.write(f"VERTEX_SE3:QUAT {i} " + _fmt(t) + " " + _fmt(q[[1, 2, 3, 0]]) + "\n")

    def get_batch_dataset(self, batch_idx: int = 0) -> "PoseGraphDataset":
        assert batch_idx < self.num_batches
        start = batch_idx * self.batch_size
        end = min(start + self.batch_size, self.dataset_size)
        group_cls = self.poses[0].__class__

        def cut(v):
            return group_cls(tensor=v.tensor[start:end].clone(), name=v.name + "__batch")
        poses = [cut(p) for p in self.poses]
        gt_poses = None if self.gt_poses is None else [cut(p) for p in self.gt_poses]
        edges = [PoseGraphEdge(e.i, e.j, relative_pose=cut(e.relative_pose), weight=e.weight) for e in self.edges]
        return PoseGraphDataset(poses, edges, gt_poses, batch_size=self.batch_size)

    def to(self, *args, **kwargs):
        for group in (self.gt_poses, self.poses, self.edges):
            if group is not None:
                for item in group:
                    item.to(*args, **kwargs)


def pose_graph_objective(verts: List[Pose], edges: List[PoseGraphEdge], dtype: torch.dtype = torch.float64,
                         prior_scale: float = 1e-6) -> th.Objective:
    """The objective of examples/pose_graph/pose_graph_benchmark.py:45-63: one Between per edge with the edge's own
    DiagonalCostWeight, and a Difference prior pinning pose 0 to its initial value with ScaleCostWeight(prior_scale)."""
    objective = th.Objective(dtype)
    for edge in edges:
        objective.add(th.Between(verts[edge.i], verts[edge.j], edge.relative_pose, edge.weight))
    target = verts[0].__class__(tensor=verts[0].tensor.clone(), name=verts[0].name + "PRIOR")
    objective.add(th.Difference(verts[0], target, th.ScaleCostWeight(torch.tensor(prior_scale, dtype=dtype, device=verts[0].device))))
    return objective
