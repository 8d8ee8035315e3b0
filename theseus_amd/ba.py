"""Bundle adjustment on the HIP back end (BASELINE.json configs[3]; examples/bundle_adjustment.py:103-160).

Objectives whose optimisation variables are camera poses (SE3) and world points (Point3), with Reprojection costs
(optionally wrapped in RobustCostFunction) and Difference priors on cameras / points.  The damped normal equations the
reference forms with DenseLinearization + CholeskyDenseSolver (dense n x n, n = 6C + 3Np: 27648^2 at configs[3]) are solved
EXACTLY by eliminating the points: fused block assembly (thx_ba_assemble), Schur complement on the camera block
(thx_ba_schur), the tiled dense Cholesky of the (6C)^2 reduced system (thx_chol_factor_forward / thx_chol_solve_backward),
back substitution (thx_ba_backsub).  Column order of this linearization: cameras, then points, each in objective order --
a ``VariableOrdering`` like any user-supplied one (theseus/optimizer/linearization.py:18-41).
"""
import contextlib
import dataclasses
from typing import Any, Dict, Optional, Type, Union

import numpy as np
import torch

from . import _lib
from .core import Objective, Variable
from .compiler import PoseGraphStructure
from .kernels import PGTensors, default_kernels, fast_approx_local_jacobians, round_up
from .linear_solver import LinearSolver
from .linearization import Linearization, VariableOrdering
from .packed import (UnsupportedObjective, _AuxDeepStamp, _aux_vars, _kind, _opt_deep_stamp, _radius_vars, _unwrap_robust,
                     _views_deep_stamp, _weight_diag)

ERR_CHUNKS = _lib.THX_BA_ERR_CHUNKS


import operator

_GET_TENSOR = {True: operator.attrgetter("_tensor"), False: operator.attrgetter("tensor")}   # own Variable: skip the property
_DATA_PTR, _VERSION, _NUM_UPDATES = operator.methodcaller("data_ptr"), operator.attrgetter("_version"), operator.attrgetter("_num_updates")

def _csr(owner: np.ndarray, n: int, secondary: Optional[np.ndarray] = None):
    """ids grouped by owner (stable / sorted by ``secondary``) -> (ptr (n+1), ids)."""
    ids = np.arange(owner.shape[0])
    order = np.lexsort((ids, secondary, owner)) if secondary is not None else np.argsort(owner, kind="stable")
    ptr = np.zeros(n + 1, np.int64)
    np.add.at(ptr, owner + 1, 1)
    return np.cumsum(ptr), ids[order]


class BAStructure:
    """Immutable topology of a bundle-adjustment objective + the index tables of the kernels (host, numpy)."""

    def __init__(self, num_cams: int, num_points: int, obs_cam, obs_pt, cam_prior_cam, pt_prior_pt):
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
        self.num_cams, self.num_points = int(num_cams), int(num_points)
        oc, op = np.asarray(obs_cam, np.int64).reshape(-1), np.asarray(obs_pt, np.int64).reshape(-1)
        cpc, ppp = np.asarray(cam_prior_cam, np.int64).reshape(-1), np.asarray(pt_prior_pt, np.int64).reshape(-1)
        if oc.size and (oc.min() < 0 or oc.max() >= num_cams or op.min() < 0 or op.max() >= num_points):
            raise ValueError("observation index out of range")
        self.num_obs, self.num_cam_priors, self.num_pt_priors = oc.size, cpc.size, ppp.size
        pt_ptr, pt_obs = _csr(op, num_points, oc)
        cam_ptr, cam_obs = _csr(oc, num_cams, op)
        cpp, cpi = _csr(cpc, num_cams)
        ppptr, ppi = _csr(ppp, num_points)
        # Schur pairs: (o1, o2) observe the same point and cam(o2) <= cam(o1); grouped by cam(o1), sorted by cam(o2)
        o1s, o2s = [], []
        for p in range(num_points):
            obs = pt_obs[pt_ptr[p]:pt_ptr[p + 1]]
            if obs.size:
                a, b = np.meshgrid(obs, obs, indexing="ij")
                keep = oc[b] <= oc[a]
                o1s.append(a[keep])
                o2s.append(b[keep])
        o1 = np.concatenate(o1s) if o1s else np.zeros(0, np.int64)
        o2 = np.concatenate(o2s) if o2s else np.zeros(0, np.int64)
        order = np.lexsort((o2, o1, oc[o2], oc[o1]))
        o1, o2 = o1[order], o2[order]
        pair_ptr = np.zeros(num_cams + 1, np.int64)
        np.add.at(pair_ptr, oc[o1] + 1, 1)
        self.num_pairs = o1.size
        # off-diagonal blocks (c1, c2 < c1) of the reduced system: runs of the sorted pair list (a camera's diagonal pairs,
        # c2 = c1, close its range, so the runs are given as [begin, end) and not as one CSR array)
        pc1, pc2 = oc[o1], oc[o2]
        ndiag = np.zeros(num_cams, np.int64)
        np.add.at(ndiag, pc1[pc1 == pc2], 1)
        pair_dptr = np.cumsum(pair_ptr)[1:] - ndiag
        off = np.flatnonzero(pc2 < pc1)
        if off.size:
            start = np.ones(off.size, bool)
            start[1:] = (pc1[off][1:] != pc1[off][:-1]) | (pc2[off][1:] != pc2[off][:-1])
            starts = np.flatnonzero(start)
            blk_begin = off[starts]
            blk_end = off[np.append(starts[1:] - 1, off.size - 1)] + 1
            blk_c1, blk_c2 = pc1[blk_begin], pc2[blk_begin]
        else:
            blk_begin = blk_end = blk_c1 = blk_c2 = np.zeros(0, np.int64)
        self.num_blocks = int(blk_begin.size)
        self.t = dict(obs_cam=i32(oc), obs_pt=i32(op), pt_ptr=i32(pt_ptr), pt_obs=i32(pt_obs), cam_ptr=i32(cam_ptr),
                      cam_obs=i32(cam_obs), cam_prior_cam=i32(cpc), cam_prior_ptr=i32(cpp), cam_prior_id=i32(cpi),
                      pt_prior_pt=i32(ppp), pt_prior_ptr=i32(ppptr), pt_prior_id=i32(ppi), pair_ptr=i32(np.cumsum(pair_ptr)),
                      pair_o1=i32(o1), pair_o2=i32(o2), pair_c2=i32(oc[o2]), pair_dptr=i32(pair_dptr),
                      blk_ptr=i32(np.stack([blk_begin, blk_end], 1).reshape(-1)), blk_c1=i32(blk_c1), blk_c2=i32(blk_c2))
        self._dev: Dict[str, "DeviceBA"] = {}

    @property
    def n(self) -> int:
        return 6 * self.num_cams + 3 * self.num_points

    def on(self, device) -> "DeviceBA":
        key = str(device)
        if key not in self._dev:
            self._dev[key] = DeviceBA(self, device)
        return self._dev[key]


class DeviceBA:
    def __init__(self, s: BAStructure, device):
        self.host = s
        self.t = {k: torch.from_numpy((v if v.size else np.zeros(1, np.int32)).copy()).to(device) for k, v in s.t.items()}
        c = _lib.BAStructure()
        for k in ("num_cams", "num_points", "num_obs", "num_cam_priors", "num_pt_priors", "num_pairs", "num_blocks"):
            setattr(c, k, getattr(s, k))
        for k, v in self.t.items():
            setattr(c, k, v.data_ptr())
        self.c = c


@dataclasses.dataclass
class BATensors:
    cams: torch.Tensor              # (C, B, 3, 4)
    points: torch.Tensor            # (Np, B, 3)
    feat: torch.Tensor              # (O, 1|B, 2)
    w_obs: torch.Tensor             # (O, 1|B, 2)
    focal: torch.Tensor             # (C, 1|B, 1)
    k1: torch.Tensor
    k2: torch.Tensor
    cam_prior_target: torch.Tensor  # (Kc, 1|B, 3, 4)
    w_cam_prior: torch.Tensor       # (Kc, 1|B, 6)
    pt_prior_target: torch.Tensor   # (Kp, 1|B, 3)
    w_pt_prior: torch.Tensor        # (Kp, 1|B, 3)
    robust_obs: int = 0
    log_radius_obs: Optional[torch.Tensor] = None   # (O, 1|B, 1)

    @property
    def batch(self):
        return self.cams.shape[1]

    def c_struct(self, cams=None, points=None) -> _lib.BAData:
        cams = self.cams if cams is None else cams
        points = self.points if points is None else points
        B = cams.shape[1]
        d = _lib.BAData()
        d.batch = B
        d.cams, d.points = _lib.ptr(cams, "cams").value, _lib.ptr(points, "points").value

        def put(name, t, width, stride_name=None):
            nb = t.shape[1]
            if nb not in (1, B):
                raise ValueError(f"{name}: batch dimension {nb} is neither 1 nor {B}")
            setattr(d, name, _lib.ptr(t, name).value if t.numel() else None)
            setattr(d, stride_name or name + "_bstride", width if nb == B else 0)
        put("feat", self.feat, 2)
        put("w_obs", self.w_obs, 2)
        if len({self.focal.shape[1], self.k1.shape[1], self.k2.shape[1]}) != 1:
            raise ValueError("focal_length / calib_k1 / calib_k2 must share their batch size")
        for name in ("focal", "k1", "k2"):
            put(name, getattr(self, name), 1, "calib_bstride")
        d.robust_obs = int(self.robust_obs)
        if self.robust_obs:
            put("log_radius_obs", self.log_radius_obs, 1)
        put("cam_prior_target", self.cam_prior_target, 12)
        put("w_cam_prior", self.w_cam_prior, 6)
        put("pt_prior_target", self.pt_prior_target, 3)
        put("w_pt_prior", self.w_pt_prior, 3)
        return d


class PackedBA:
    """Packed device representation of a bundle-adjustment objective (the PackedPoseGraph of this problem class)."""

    group = "BA"

    def __init__(self, objective: Objective, kernels=None):
        self.objective = objective
        self.K = kernels or default_kernels()
        self.cam_vars, self.pt_vars = [], []
        for v in objective.optim_vars.values():
            k = _kind(v)
            if k == "SE3":
                self.cam_vars.append(v)
            elif "Point3" in {c.__name__ for c in type(v).__mro__}:
                self.pt_vars.append(v)
            else:
                raise UnsupportedObjective(f"HIP bundle adjustment optimises SE3 cameras and Point3 points; got "
                                           f"{type(v).__name__} ({v.name}).  There is no CPU/eager fallback.")
        if not self.cam_vars or not self.pt_vars:
            raise UnsupportedObjective("HIP bundle adjustment needs at least one SE3 camera and one Point3 point.")
        ci = {v.name: k for k, v in enumerate(self.cam_vars)}
        pi = {v.name: k for k, v in enumerate(self.pt_vars)}
        self.obs_costs, self.obs_radius, self.cam_prior_costs, self.pt_prior_costs = [], [], [], []
        self.cc_costs, cc_edges = [], []   # camera-camera Between costs (odometry): a pose graph over the cameras
        obs_cam, obs_pt, cpc, ppp, kinds = [], [], [], [], set()
        row, rows = 0, ([], [], [], [])   # first row of every cost in the reference's (B, m) layouts: cost ADD order
        for wrapped in objective.cost_functions.values():
            c, loss, radius = _unwrap_robust(wrapped)
            names = {k.__name__ for k in type(c).__mro__}
            if "Reprojection" in names:
                obs_cam.append(ci[c.camera_pose.name])
                obs_pt.append(pi[c.world_point.name])
                self.obs_costs.append(c)
                self.obs_radius.append(radius)
                kinds.add(loss)
                rows[0].append(row)
                row += 2
            elif _kind(c) == "Difference" and not loss:
                if c.var.name in ci:
                    cpc.append(ci[c.var.name])
                    self.cam_prior_costs.append(c)
                    rows[1].append(row)
                    row += 6
                else:
                    ppp.append(pi[c.var.name])
                    self.pt_prior_costs.append(c)
                    rows[2].append(row)
                    row += 3
            elif _kind(c) == "Between" and not loss and c.v0.name in ci and c.v1.name in ci:
                cc_edges.append((ci[c.v0.name], ci[c.v1.name]))
                self.cc_costs.append(c)
                rows[3].append(row)
                row += 6
            else:
                raise UnsupportedObjective(f"HIP bundle adjustment has no fused kernel for {type(wrapped).__name__} "
                                           f"({wrapped.name}); supported: Reprojection (optionally robust), Difference, "
                                           "Between on two cameras.")
        if len(kinds) > 1:
            raise UnsupportedObjective("HIP bundle adjustment: all Reprojection costs must share one robust loss kind.")
        self.robust_obs = kinds.pop() if kinds else _lib.LOSS_NONE
        self._refuse_fast_approx(UnsupportedObjective)
        self.structure = BAStructure(len(self.cam_vars), len(self.pt_vars), obs_cam, obs_pt, cpc, ppp)
        # Camera-camera costs ride on the pose-graph kernels (thx_pg_assemble / thx_pg_error / thx_pg_jacobians / thx_pg_vjp over
        # the CAMERA buffer): their 6 x 6 blocks are added to the camera part of the system before / after the point elimination
        # (HipSchurLinearizationCore._assemble, HipSchurSolverCore._solve).
        self.cc_structure = PoseGraphStructure.build(len(self.cam_vars), cc_edges, [], edge_row_start=rows[3]) if cc_edges else None
        self.cc_tensors: Optional[PGTensors] = None
        self.n = self.structure.n
        self.nc = 6 * len(self.cam_vars)
        self.m = objective.dim()
        self.cost_rows = tuple(np.asarray(r if r else [0], dtype=np.int32) for r in rows)
        self._cost_rows_dev: Dict[str, Any] = {}
        self.version = objective.current_version
        self.tensors: Optional[BATensors] = None
        seen, self._tracked_list = set(), []          # shared calibration / weights / radius appear once
        for v in self._walk_tracked():
            if id(v) not in seen:
                seen.add(id(v))
                self._tracked_list.append(v)
        self._own_variables = all(isinstance(v, Variable) for v in self._tracked_list)
        self._stamp = None
        self._deep_stamp = None
        self._aux_stamp = _AuxDeepStamp()
        self._keep_graph_tensors = False   # (see PackedPoseGraph)
        self._global_stamp = -1
        self._vars_stale = False
        self._state_exposed = False
        self._scratch = {}

    def _refuse_fast_approx(self, exc=NotImplementedError):
        """See PackedPoseGraph._refuse_fast_approx (the SE3 camera priors would silently get their exact Jacobian)."""
        if self.cam_prior_costs and fast_approx_local_jacobians():
            raise exc("HIP bundle adjustment: the global option fast_approx_local_jacobians=True is not fused into the "
                      "kernels (camera Difference/Local costs); there is no CPU/eager fallback.")

    # ---- packing ------------------------------------------------------------------------------------------
    def _tracked(self):
        return self._tracked_list

    def tracked_list(self):
        return self._tracked_list

    def _walk_tracked(self):
        yield from self.cam_vars
        yield from self.pt_vars
        for c, r in zip(self.obs_costs, self.obs_radius):
            yield c.image_feature_point
            yield c.focal_length
            yield c.calib_k1
            yield c.calib_k2
            yield from _aux_vars(c.weight)
            yield from _radius_vars(r)
        for c in self.cam_prior_costs + self.pt_prior_costs:
            yield c.target
            yield from _aux_vars(c.weight)
        for c in self.cc_costs:
            yield c.measurement
            yield from _aux_vars(c.weight)

    def _counters_unchanged(self) -> bool:
        return self._own_variables and self._stamp is not None and Variable._global_updates == self._global_stamp

    def _current_stamp(self, deep: bool = False, count: Optional[int] = None):
        # the objective is frozen once an optimizer holds it (Optimizer.optimize checks its version): the walk over the cost
        # functions is done once, later stamps are one pass over the cached list (7 k variables at the headline size)
        tracked = self.__dict__.get("_tracked_list")
        if tracked is None:
            tracked = self._tracked_list = list(self._tracked())
        if count is not None:
            tracked = tracked[:count]
        if deep:
            # (optimisation part, auxiliary part): see PackedPoseGraph._current_stamp / packed._AuxDeepStamp
            ts = list(map(_GET_TENSOR[self._own_variables], tracked))
            n_opt = min(len(self.cam_vars) + len(self.pt_vars), len(ts))
            opt = _opt_deep_stamp(ts[:n_opt])
            return opt if count is not None else (opt, self._aux_stamp.stamp(ts[n_opt:]))
        return tuple(map(_NUM_UPDATES, tracked))

    @staticmethod
    def _stack(ts, B):
        if all(t.shape[0] == ts[0].shape[0] for t in ts):
            return torch.stack(ts, dim=0).contiguous()
        return torch.stack([t.expand(B, *t.shape[1:]) for t in ts], dim=0).contiguous()

    @contextlib.contextmanager
    def pinned(self, tensors, cc_tensors=None):
        """Run the kernels on ANOTHER snapshot of the problem (a saved iterate, in a backward pass): ``tensors`` / ``cc_tensors`` stand
        in for the live ones and sync() leaves them alone."""
        live, live_cc, was = self.tensors, self.cc_tensors, self.__dict__.get("_pinned", False)
        self.tensors, self.cc_tensors, self._pinned = tensors, (cc_tensors if cc_tensors is not None else live_cc), True
        try:
            yield
        finally:
            self.tensors, self.cc_tensors, self._pinned = live, live_cc, was

    def sync(self, force: bool = False, deep: bool = False):
        if self.__dict__.get("_pinned", False):
            return
        deep = deep or not self._own_variables   # see PackedPoseGraph.sync
        if (not force and not deep and self.tensors is not None
                and Variable._global_updates == self._global_stamp):
            return
        # (nobody called Variable.update() / to() since the last look: the update counters -- the shallow stamp -- are what they
        #  were; a pass over 42 k variables of a bundle-adjustment objective is ~10 ms of host time per optimize())
        shallow = self._own_variables or self._stamp is None    # (reference Variables: the deep stamp alone, see PackedPoseGraph.sync)
        stamp = self._stamp if (self._counters_unchanged() or not shallow) else self._current_stamp()
        dstamp = self._current_stamp(deep=True) if deep else None
        if (not force and self.tensors is not None and stamp == self._stamp and (not deep or dstamp == self._deep_stamp)):
            self._global_stamp = Variable._global_updates
            return
        self._deep_stamp = dstamp if dstamp is not None else self._current_stamp(deep=True)
        self.flush_variables()
        obj = self.objective
        obj._resolve_batch_size()
        B = obj.batch_size
        dev, dt = self.cam_vars[0].device, obj.dtype
        full = lambda v, *s: v.tensor.expand(B, *s) if v.shape[0] != B else v.tensor  # noqa: E731
        cams = self._stack([full(v, 3, 4) for v in self.cam_vars], B)
        pts = self._stack([full(v, 3) for v in self.pt_vars], B)
        empty = lambda *s: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
        cam_of = {}
        for c in self.obs_costs:  # calibration is read per camera: all observations of a camera must share it
            k = c.camera_pose.name
            trip = (c.focal_length, c.calib_k1, c.calib_k2)
            if k in cam_of and any(a is not b for a, b in zip(cam_of[k], trip)):
                raise UnsupportedObjective("HIP bundle adjustment: the Reprojection costs of one camera must share their "
                                           "focal_length / calib_k1 / calib_k2 variables.")
            cam_of[k] = trip
        one = lambda: torch.ones(1, 1, dtype=dt, device=dev)  # noqa: E731
        zero = lambda: torch.zeros(1, 1, dtype=dt, device=dev)  # noqa: E731
        calib = [cam_of.get(v.name) for v in self.cam_vars]
        focal = self._stack([c[0].tensor.view(-1, 1) if c else one() for c in calib], B)
        k1 = self._stack([c[1].tensor.view(-1, 1) if c else zero() for c in calib], B)
        k2 = self._stack([c[2].tensor.view(-1, 1) if c else zero() for c in calib], B)
        nb = max(focal.shape[1], k1.shape[1], k2.shape[1])
        focal, k1, k2 = (t.expand(-1, nb, 1).contiguous() for t in (focal, k1, k2))
        O, Kc, Kp = len(self.obs_costs), len(self.cam_prior_costs), len(self.pt_prior_costs)
        feat = self._stack([c.image_feature_point.tensor for c in self.obs_costs], B) if O else empty(0, 1, 2)
        w_obs = self._stack([_weight_diag(c.weight, 2) for c in self.obs_costs], B) if O else empty(0, 1, 2)
        lr = self._stack([r.tensor.view(-1, 1) for r in self.obs_radius], B) if self.robust_obs else None
        cpt = self._stack([c.target.tensor for c in self.cam_prior_costs], B) if Kc else empty(0, 1, 3, 4)
        wcp = self._stack([_weight_diag(c.weight, 6) for c in self.cam_prior_costs], B) if Kc else empty(0, 1, 6)
        ppt = self._stack([c.target.tensor for c in self.pt_prior_costs], B) if Kp else empty(0, 1, 3)
        wpp = self._stack([_weight_diag(c.weight, 3) for c in self.pt_prior_costs], B) if Kp else empty(0, 1, 3)
        self.tensors = BATensors(cams=cams, points=pts, feat=feat, w_obs=w_obs, focal=focal, k1=k1, k2=k2,
                                 cam_prior_target=cpt, w_cam_prior=wcp, pt_prior_target=ppt, w_pt_prior=wpp,
                                 robust_obs=self.robust_obs, log_radius_obs=lr)
        if self.cc_costs:
            self.cc_tensors = PGTensors(poses=cams, meas=self._stack([c.measurement.tensor for c in self.cc_costs], B),
                                        w_between=self._stack([_weight_diag(c.weight, 6) for c in self.cc_costs], B),
                                        prior_target=empty(0, 1, 3, 4), w_prior=empty(0, 1, 6))
        if (not (cams.requires_grad or pts.requires_grad) and (self._keep_graph_tensors or not self._own_variables)
                and any(v.tensor.requires_grad for v in self.cam_vars + self.pt_vars)):
            # (packed under no_grad from tensors that carry autograd history: see PackedPoseGraph.sync -- the variables keep them)
            self._stamp = self._current_stamp() if not self._counters_unchanged() else self._stamp
            self._global_stamp = Variable._global_updates
            self._vars_stale = False
            self._state_exposed = False
        else:
            self._repoint_variables()

    def _repoint_variables(self):
        # (after the implicit last step the state carries an autograd graph: the per-variable views stay attached to it)
        with torch.set_grad_enabled(self.tensors.cams.requires_grad or self.tensors.points.requires_grad):
            attr = "_tensor" if self._own_variables else "tensor"   # (own Variable: no update count for a re-pointing, core.py)
            for v, t in zip(self.cam_vars, self.tensors.cams.unbind(0)):  # one call builds all the views
                setattr(v, attr, t)
            for v, t in zip(self.pt_vars, self.tensors.points.unbind(0)):
                setattr(v, attr, t)
        if not self._counters_unchanged():
            self._stamp = self._current_stamp()
        # deep stamp: only the optimisation variables were re-pointed here -- the auxiliary variables' entries (the bulk: 1 k
        # measurements of a pose graph, 33 k features of a bundle-adjustment problem) are still the ones sync() looked at
        n_opt = len(self.cam_vars) + len(self.pt_vars)
        old = self._deep_stamp
        if old is not None:
            self._deep_stamp = (_views_deep_stamp((self.tensors.cams, self.tensors.points)), old[1])
        else:
            self._deep_stamp = self._current_stamp(deep=True)
        self._global_stamp = Variable._global_updates
        self._vars_stale = False
        self._state_exposed = True

    def privatize_state(self):
        """See PackedPoseGraph.privatize_state: never recycle a state buffer the variables' tensors view."""
        if self._state_exposed:
            with torch.no_grad():
                self.tensors.cams = self.tensors.cams.detach().clone()
                self.tensors.points = self.tensors.points.detach().clone()
            self._state_exposed = False
            self._vars_stale = True

    def flush_variables(self):
        if self._vars_stale and self.tensors is not None:
            self._repoint_variables()

    # ---- optimisation state = (cams, points) ------------------------------------------------------------------
    @property
    def state(self):
        return (self.tensors.cams, self.tensors.points)

    @property
    def device(self):
        return self.tensors.cams.device

    @property
    def batch(self):
        return self.tensors.batch

    @property
    def optim_variables(self):
        return self.cam_vars + self.pt_vars

    @property
    def dstruct(self):
        return self.structure.on(self.tensors.cams.device)

    def alloc_state(self):
        return (torch.empty_like(self.tensors.cams), torch.empty_like(self.tensors.points))

    def clone_state(self):
        return (self.tensors.cams.clone(), self.tensors.points.clone())

    def swap_state(self, new, repoint: bool = False):
        old = self.state
        self.tensors.cams, self.tensors.points = new
        if repoint:
            self._repoint_variables()
        else:
            self._vars_stale = True
        return old

    # ---- BackwardMode.UNROLL / TRUNCATED (theseus_amd/nonlinear.py: the differentiable tail loop) ------------------------------
    def prepare_unroll(self):
        """Before the differentiable tail loop: re-pack the auxiliary tensors WITH their autograd history (once)."""
        self.flush_variables()
        self.sync(force=True)

    def unrolled_step(self, opt, X, frozen: Optional[torch.Tensor], kwargs):
        """(cams, points) -> ((cams exp(step * delta_c), points + step * delta_p) where not ``frozen``, delta) as ONE autograd node."""
        t = self.tensors
        cc = (self.cc_tensors.meas, self.cc_tensors.w_between) if self.cc_costs else ()
        cams, pts, delta = BAUnrolledIteration.apply(opt, self, frozen, kwargs, X[0], X[1], t.feat, t.w_obs, t.focal, t.k1, t.k2,
                                                     t.log_radius_obs, t.cam_prior_target, t.w_cam_prior, t.pt_prior_target,
                                                     t.w_pt_prior, *cc)
        return (cams, pts), delta

    def state_with_graph(self, tensors):
        """(cams, points) assembled from the variables' OWN tensors (``optim_variables`` order: cameras, then points), autograd
        history kept: where BackwardMode.UNROLL starts, so that gradients reach the values the caller passed in."""
        B, C = self.batch, len(self.cam_vars)
        full = [t if t.shape[0] == B else t.expand(B, *t.shape[1:]) for t in tensors]
        return (torch.stack(full[:C], dim=0), torch.stack(full[C:], dim=0))

    def where_state(self, mask: torch.Tensor, a, b):
        """Per problem: ``a`` where ``mask`` else ``b`` (differentiable torch select on both halves of the state)."""
        return (torch.where(mask.view(1, -1, 1, 1), a[0], b[0]), torch.where(mask.view(1, -1, 1), a[1], b[1]))

    def keep_where(self, mask, out):
        self.copy_where(mask, self.state, out)

    def copy_where(self, mask, src, dst):
        self.K.copy_where(mask, src[0], dst[0])
        self.K.copy_where(mask, src[1], dst[1])

    def history_dict(self, hist, dtype):
        hc, hp = hist[0].to(dtype).cpu(), hist[1].to(dtype).cpu()
        out = {v.name: hc[:, k].movedim(0, -1).contiguous() for k, v in enumerate(self.cam_vars)}
        out.update({v.name: hp[:, k].movedim(0, -1).contiguous() for k, v in enumerate(self.pt_vars)})
        return out

    def solution_dict(self, state):
        out = {v.name: state[0][k].cpu() for k, v in enumerate(self.cam_vars)}
        out.update({v.name: state[1][k].cpu() for k, v in enumerate(self.pt_vars)})
        return out

    # ---- fused operations ----------------------------------------------------------------------------------------
    def _buf(self, key, shape, dtype=None):
        t = self._scratch.get(key)
        dt = dtype or self.objective.dtype
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dt or t.device != self.device:
            t = torch.empty(*shape, dtype=dt, device=self.device)
            self._scratch[key] = t
        return t

    def error_metric(self, state=None, out: Optional[torch.Tensor] = None, poses=None):
        self.sync()
        B = self.batch
        part = self._buf("err_part", (ERR_CHUNKS, B))
        err = out if out is not None else torch.empty(B, dtype=self.objective.dtype, device=part.device)
        cams, pts = state if state is not None else (None, None)
        self.K.ba_error(self.dstruct, self.tensors, part, err, cams=cams, points=pts)
        if self.cc_costs:   # + the camera-camera costs' share of the metric (thx_pg_error over the camera buffer)
            err_cc = self._buf("err_cc", (B,))
            self.K.pg_error(self.cc_dstruct, self.cc_tensors, self._buf("err_part_cc", (_lib.THX_ERR_CHUNKS, B)), err_cc,
                            poses=cams if cams is not None else self.tensors.cams)
            err.add_(err_cc)
        return err

    @property
    def cc_dstruct(self):
        return self.cc_structure.on(self.device)

    def retract(self, delta: torch.Tensor, step: float, ignore_mask: Optional[torch.Tensor], out):
        self.sync()
        m = None
        if ignore_mask is not None:
            m = ignore_mask if ignore_mask.dtype == torch.uint8 else ignore_mask.to(torch.uint8)
        self.K.se3_retract(self.tensors.cams, delta, step, m, out[0])   # columns [0, 6C)
        self.K.vec_retract(self.tensors.points, delta, self.nc, step, m, out[1])
        return out

    def error_vector(self):
        raise NotImplementedError("Objective.error() of a bundle-adjustment objective is not materialised on the HIP back end "
                                  "(use error_metric())")


class HipSchurLinearizationCore:
    """Back-end half of the bundle-adjustment linearization, independent of which ``Linearization`` ABC it is mixed into
    (theseus_amd's mirror below, the real ``theseus.optimizer.Linearization`` in theseus_amd/plugin.py): block form
    (Hcc, Hpp, Hcp) + g + diag(H), never the dense H.  ``ordering``: cameras, then points."""

    @staticmethod
    def _schur_setup(objective, ordering, kernels, ordering_cls):
        packed = getattr(objective, "_packed", None)
        if not isinstance(packed, PackedBA) or packed.version != objective.current_version or (
                kernels is not None and packed.K is not kernels):
            packed = PackedBA(objective, kernels)
            objective._packed = packed
        if ordering is not None:
            raise NotImplementedError("HipSchurLinearization fixes the variable ordering: cameras, then points.")
        ordering = ordering_cls(objective, default_order=False)
        for v in packed.cam_vars + packed.pt_vars:
            ordering.append(v)
        return packed, ordering

    def _schur_init(self, packed):
        self.packed, self.K = packed, packed.K
        self.Hcc = self.Hpp = self.W = self.gd = self.g = self.diag = None

    @property
    def n(self):
        return self.packed.n

    def _ensure_buffers(self):
        p = self.packed
        p.sync()
        B, s = p.batch, p.structure
        dev, dt = p.device, self.objective.dtype
        if self.g is None or self.g.shape[0] != B or self.g.device != dev or self.g.dtype != dt:
            new = lambda *sh: torch.empty(*sh, dtype=dt, device=dev)  # noqa: E731
            f64 = lambda *sh: torch.empty(*sh, dtype=torch.float64, device=dev)  # noqa: E731
            # block quantities in fp64 for every dtype: the Schur complement cancels at the scale of Hcc
            # (planar workspaces: (entity, component, B), the batch innermost -- include/theseus_hip.h, "Layouts")
            self.Hcc, self.Hpp = f64(s.num_cams, 36, B), f64(s.num_points, 6, B)
            self.W, self.gd = f64(max(s.num_obs, 1), 18, B), f64(B, p.n)
            self.g, self.diag = new(B, p.n), new(B, p.n)
            if p.cc_costs:   # index tables of the camera-camera costs' blocks (their Jacobians: thx_pg_jacobians, per linearize)
                st = p.cc_structure
                self._cc_i = torch.from_numpy(st.edge_i.astype(np.int64)).to(dev)
                self._cc_j = torch.from_numpy(st.edge_j.astype(np.int64)).to(dev)
                hi, lo = torch.maximum(self._cc_i, self._cc_j), torch.minimum(self._cc_i, self._cc_j)
                k = torch.arange(6, device=dev)
                self._cc_rows = (6 * hi.view(-1, 1, 1) + k.view(1, 6, 1)).expand(-1, 6, 6)     # (E, 6, 6) rows / columns of the
                self._cc_cols = (6 * lo.view(-1, 1, 1) + k.view(1, 1, 6)).expand(-1, 6, 6)     # off-diagonal block of edge e in S
                self._cc_v0_is_row = (self._cc_i > self._cc_j).view(-1, 1, 1, 1)

    def _assemble(self):
        self._ensure_buffers()
        p = self.packed
        p._refuse_fast_approx()
        self.K.ba_assemble(p.dstruct, p.tensors, self.Hcc, self.Hpp, self.W, self.gd, self.g, self.diag)
        if p.cc_costs:
            # camera-camera Between costs: their weighted Jacobian blocks from thx_pg_jacobians over the camera buffer; the 6 x 6
            # products are a handful of small batched GEMMs.  Diagonal blocks and gradient join the camera blocks BEFORE the point
            # elimination (damping sees them); the off-diagonal blocks are added to the reduced system AFTER it
            # (HipSchurSolverCore._solve):  S = (Hcc + Hodo)' - Hcp Hpp'^-1 Hpc.
            E, B, dt, dev = len(p.cc_costs), self.g.shape[0], self.g.dtype, self.g.device
            J0, J1 = (torch.empty(E, B, 6, 6, dtype=dt, device=dev) for _ in range(2))
            eb = torch.empty(E, B, 6, dtype=dt, device=dev)
            self.K.pg_jacobians(p.cc_dstruct, p.cc_tensors, J0, J1, eb, torch.empty(1, B, 6, 6, dtype=dt, device=dev),
                                torch.empty(1, B, 6, dtype=dt, device=dev), poses=p.tensors.cams)
            self._cc_J = (J0, J1)                                                  # (Av reads them)
            J0d, J1d, ed = J0.double(), J1.double(), eb.double().unsqueeze(3)
            D0, D1 = J0d.transpose(2, 3) @ J0d, J1d.transpose(2, 3) @ J1d            # (E, B, 6, 6) diagonal-block contributions
            g0, g1 = -(J0d.transpose(2, 3) @ ed).squeeze(3), -(J1d.transpose(2, 3) @ ed).squeeze(3)
            planar = lambda x: x.reshape(E, B, -1).transpose(1, 2)                 # noqa: E731  (E, B, ...) -> (E, comps, B)
            self.Hcc.index_add_(0, self._cc_i, planar(D0)).index_add_(0, self._cc_j, planar(D1))
            nc = p.nc
            gsum = torch.zeros(B, p.structure.num_cams, 6, dtype=torch.float64, device=dev)
            gsum.index_add_(1, self._cc_i, g0.transpose(0, 1)).index_add_(1, self._cc_j, g1.transpose(0, 1))
            dsum = torch.zeros_like(gsum)
            dsum.index_add_(1, self._cc_i, D0.diagonal(dim1=2, dim2=3).transpose(0, 1))
            dsum.index_add_(1, self._cc_j, D1.diagonal(dim1=2, dim2=3).transpose(0, 1))
            self.gd[:, :nc].add_(gsum.reshape(B, nc))
            self.g[:, :nc].add_(gsum.reshape(B, nc).to(dt))
            self.diag[:, :nc].add_(dsum.reshape(B, nc).to(dt))
            # block (max(i, j), min(i, j)) of S: J_row^T J_col
            off = torch.where(self._cc_v0_is_row, J0d.transpose(2, 3) @ J1d, J1d.transpose(2, 3) @ J0d)
            self._cc_off = off.transpose(0, 1).to(dt).contiguous()                 # (B, E, 6, 6)

    def _linearize_jacobian_impl(self):
        raise NotImplementedError("the dense Jacobian of a bundle-adjustment objective is not materialised")

    def _linearize_hessian_impl(self, _detach_hessian: bool = False):
        self._assemble()

    def _ata_impl(self) -> torch.Tensor:
        raise NotImplementedError("the dense Hessian of a bundle-adjustment objective is not materialised "
                                  "(Hcc / Hpp / W hold its blocks)")

    def _atb_impl(self) -> torch.Tensor:
        return self.g.unsqueeze(2)

    def Av(self, v: torch.Tensor) -> torch.Tensor:
        """(B, n) -> (B, m), dense_linearization.py:73-74: per-cost Jacobian blocks recomputed by thx_ba_av at the variables'
        current values (those of this linearization wherever the reference's optimizers call it: dogleg.py:66,
        trust_region.py:97), never the dense Jacobian.  Rows in cost add order, columns cameras then points."""
        p = self.packed
        p.sync()
        key = str(v.device)
        if key not in p._cost_rows_dev:
            p._cost_rows_dev[key] = tuple(torch.from_numpy(r).to(v.device) for r in p.cost_rows)
        v = v.to(self.objective.dtype).contiguous()
        out_t = torch.empty(p.m, v.shape[0], dtype=v.dtype, device=v.device)
        self.K.ba_av(p.dstruct, p.tensors, v, p._cost_rows_dev[key][:3], out_t)
        out = out_t.t().contiguous()
        if p.cc_costs:   # rows of the camera-camera costs: J0 v_i + J1 v_j (the blocks of this linearization)
            B = v.shape[0]
            J0, J1 = self._cc_J
            vc = v[:, :p.nc].reshape(B, -1, 6)
            vi = vc[:, self._cc_i].transpose(0, 1).unsqueeze(3)   # (E, B, 6, 1)
            vj = vc[:, self._cc_j].transpose(0, 1).unsqueeze(3)
            rows = (p._cost_rows_dev[key][3].long().view(-1, 1) + torch.arange(6, device=v.device)).view(-1)
            out[:, rows] = (J0 @ vi + J1 @ vj).squeeze(3).transpose(0, 1).reshape(B, -1)
        return out

    def diagonal_scaling(self, v: torch.Tensor) -> torch.Tensor:
        return self.diag * v

    def lm_accept(self, delta, damping, prev_err, new_err, ellipsoidal, accept, down, up, reject):
        self.K.lm_accept_diag(delta, self.g, self.diag, self.n, damping, prev_err, new_err, ellipsoidal, accept, down, up, reject)


class HipSchurLinearization(HipSchurLinearizationCore, Linearization):
    def __init__(self, objective: Objective, ordering: Optional[VariableOrdering] = None, kernels=None, **kwargs):
        packed, ordering = self._schur_setup(objective, ordering, kernels, VariableOrdering)
        Linearization.__init__(self, objective, ordering)
        self._schur_init(packed)

    @property
    def AtA(self) -> torch.Tensor:
        return self._ata_impl()

    @property
    def Atb(self) -> torch.Tensor:
        return self._atb_impl()


class _SchurBlockList:
    """The reduced camera system as a block list (include/theseus_hip.h: thx_hblock_layout values, bd = 6): what
    theseus_amd.sparse._LevelLayout needs of a compiler.HessianBlocks -- ``blocks`` (row, col) in the solver's camera order."""

    def __init__(self, blocks: np.ndarray, num_cams: int):
        self.blocks = np.ascontiguousarray(blocks, dtype=np.int64)
        self.nblocks, self.bd, self.nvars = int(blocks.shape[0]), 6, int(num_cams)
        self.elems = self.nblocks * 36
        self.bstride = (self.elems + 3) // 4 * 4
        self._dev: Dict[str, Any] = {}

    def on(self, device):
        key = str(device)
        if key not in self._dev:
            z = torch.zeros(1, dtype=torch.int32, device=device)   # (diag_blk / inc_blk: read by the pose-graph assembly only)
            self._dev[key] = type("DeviceBlockList", (), dict(t=dict(diag_blk=z, inc_blk=z)))()
        return self._dev[key]


class HipSchurSolverCore:
    """(H + damping) delta = g of a bundle-adjustment linearization by block elimination of the points + the tiled dense
    Cholesky on the reduced camera system.  Same ``solve`` contract and failure behaviour as ``HipCholeskySolver``."""

    def _schur_solver_init(self, sparse_reduced_system: bool = True, ordering: str = "auto"):
        self.K = self.linearization.K
        self.S = self.L = self.panels = self.info_chol = self.info_pts = None
        self.pattern, self.sparse, self._want_sparse = None, False, bool(sparse_reduced_system)
        self.factor_version = 0
        self._odo_only = None
        # LEVEL MODE (round 6): S as a block list (thx_ba_schur_blocks) in a tile-level nested-dissection order of the cameras,
        # factorised along its elimination tree (thx_chol_factor_levels).  ordering: "auto" (the time model of
        # theseus_amd.sparse ranks the candidate orders at this batch size; the band order keeps the column-by-column schedule
        # on the dense frame) | "nd" / "nd<leaf>" / "md" (level mode with that order) | "natural" (rounds 3-5: dense frame).
        self.levels, self._ordering, self.ordering_info = False, str(ordering), dict(method="natural")
        self.Sc = self._level = None

    @staticmethod
    def _blocks_only_odometry(p):
        """(rows, cols) index tensors of the 6 x 6 blocks of S that ONLY camera-camera costs fill (no shared point), or None."""
        t, st = p.structure.t, p.cc_structure
        shared = set(zip(t["blk_c1"].tolist(), t["blk_c2"].tolist()))
        only = sorted({(max(i, j), min(i, j)) for i, j in zip(st.edge_i.tolist(), st.edge_j.tolist())} - shared)
        if not only:
            return None
        dev = p.device
        k = torch.arange(6, device=dev)
        c1 = 6 * torch.tensor([a for a, _ in only], device=dev).view(-1, 1, 1)
        c2 = 6 * torch.tensor([b for _, b in only], device=dev).view(-1, 1, 1)
        n = len(only)
        return (c1 + k.view(1, 6, 1)).expand(n, 6, 6), (c2 + k.view(1, 1, 6)).expand(n, 6, 6)

    def _level_setup(self, B: int) -> bool:
        """Decide the mode once (first solve): True = level mode, with the pattern / block tables built."""
        p = self.linearization.packed
        self._level = False
        if self._ordering == "natural" or not self._want_sparse or p.cc_costs or not hasattr(self.K, "ba_schur_blocks"):
            return False   # (camera-camera costs scatter-add into the dense frame: that path stays as it was)
        from .sparse import LEVEL_SCHEDULE_MAX_BATCH, TILE, LevelPattern, tile_nested_dissection
        if self._ordering == "auto" and B >= LEVEL_SCHEDULE_MAX_BATCH:
            return False   # (the batch fills the chip several times over: the column schedule's half-batch streams, sparse.level_ordering)
        s = p.structure
        C, t = s.num_cams, s.t
        if C <= TILE // 6:
            return False   # (one tile: nothing to dissect)
        c1, c2 = t["blk_c1"].astype(np.int64)[:s.num_blocks], t["blk_c2"].astype(np.int64)[:s.num_blocks]
        order, counts, info = tile_nested_dissection(C, list(zip(c1.tolist(), c2.tolist())), TILE // 6, batch_hint=int(B),
                                                     method=self._ordering)
        self.ordering_info = info
        if info["method"] == "band" and self._ordering == "auto":
            return False   # (no order beats the cameras' own at this batch size: the column-by-column schedule with its look-ahead)
        order = np.asarray(order, dtype=np.int64)
        pos = np.empty(C, dtype=np.int64)
        pos[order] = np.arange(C)
        # the block list in the SOLVER's order: block c = S_cc at (pos c, pos c); block C + k = the k-th camera pair, transposed
        # where the order puts c2 behind c1
        p1, p2 = pos[c1], pos[c2]
        blocks = np.concatenate([np.stack([pos, pos], 1), np.stack([np.maximum(p1, p2), np.minimum(p1, p2)], 1)], 0)
        self._blocklist = _SchurBlockList(blocks, C)
        self.pattern = LevelPattern(blocks, 6, counts)
        dst = (C + np.arange(c1.size, dtype=np.int64)) | ((p1 < p2).astype(np.int64) << 30)
        # vectors: the structure's camera columns <-> the padded order (LevelPattern's maps are in the solver's order)
        col = (6 * pos[:, None] + np.arange(6)[None, :]).reshape(-1)                      # solver column of structure column
        pad_of_col = self.pattern.pad_of_col[col]
        col_of_pad = np.full(self.pattern.npad, -1, dtype=np.int32)
        col_of_pad[pad_of_col] = np.arange(6 * C, dtype=np.int32)
        dev = self.linearization.g.device
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)  # noqa: E731
        self._level_t = dict(diag_blk=i32(np.arange(C)), blk_dst=i32(dst if dst.size else [0]), pad_of_col=i32(pad_of_col),
                             col_of_pad=i32(col_of_pad))
        from .sparse import _LevelLayout
        self._level_layout = _LevelLayout(self.pattern, self._blocklist, dev)
        self._level = True
        return True

    def _ensure_buffers(self):
        lin = self.linearization
        B, nc = lin.g.shape[0], lin.packed.nc
        dev, dt = lin.g.device, lin.g.dtype
        if self._level is None:
            self.levels = self._level_setup(B)
        if self.levels:
            if self.Sc is None or self.Sc.shape[0] != B or self.Sc.device != dev or self.Sc.dtype != dt:
                s, pat, T = lin.packed.structure, self.pattern, _lib.THX_TILE
                self.Sc = torch.zeros(B, self._blocklist.bstride, dtype=dt, device=dev)
                self.L = torch.zeros(B, pat.nslots, T, T, dtype=dt, device=dev)       # tile-packed factor
                self.panels = torch.empty(B, pat.ntiles, T, T, dtype=dt, device=dev)
                self.rhs, self._dc = (torch.empty(B, nc, dtype=dt, device=dev) for _ in range(2))
                self._xp, self._yp = (torch.zeros(B, pat.npad, dtype=dt, device=dev) for _ in range(2))   # (padding stays zero)
                self.Hinv = torch.empty(s.num_points, 6, B, dtype=torch.float64, device=dev)
                self.tvec = torch.empty(s.num_points, 3, B, dtype=torch.float64, device=dev)
                self.delta = torch.empty(B, lin.n, dtype=dt, device=dev)
                self.info_chol = torch.zeros(B, dtype=torch.int32, device=dev)
                self.info_pts = torch.zeros(B, dtype=torch.int32, device=dev)
                self._lam = torch.empty(B, dtype=dt, device=dev)
                self.sparse = True
            return
        if self.S is None or self.S.shape[0] != B or self.S.device != dev or self.S.dtype != dt:
            ld = round_up(nc, 32)
            nt = (nc + _lib.THX_TILE - 1) // _lib.THX_TILE
            s = lin.packed.structure
            self.S = torch.zeros(B, ld, ld, dtype=dt, device=dev)   # zero once: the block pattern is fixed
            self.L = torch.zeros_like(self.S)
            self.panels = torch.empty(B, nt, _lib.THX_TILE, _lib.THX_TILE, dtype=dt, device=dev)
            self.rhs, self._y, self._dc = (torch.empty(B, nc, dtype=dt, device=dev) for _ in range(3))
            self.Hinv = torch.empty(s.num_points, 6, B, dtype=torch.float64, device=dev)
            self.tvec = torch.empty(s.num_points, 3, B, dtype=torch.float64, device=dev)
            self.delta = torch.empty(B, lin.n, dtype=dt, device=dev)
            self.info_chol = torch.zeros(B, dtype=torch.int32, device=dev)
            self.info_pts = torch.zeros(B, dtype=torch.int32, device=dev)
            self._lam = torch.empty(B, dtype=dt, device=dev)
        if self.pattern is None:
            # Tile pattern of the reduced camera system: S has a block (c1, c2) where two cameras see a common point.  With the
            # reference's generator (cameras on a line, tracks local: data.py:311-320) S is BANDED -- at 512 cameras 158 of 300
            # lower tiles, 21 % of the dense tile products -- and the factorisation visits the structurally non-zero tiles only
            # (thx_chol_factor_sparse, bit-identical to the dense one on the same matrix).  A dense pattern keeps the dense call.
            from .sparse import TilePattern
            t = lin.packed.structure.t
            cams = np.arange(lin.packed.structure.num_cams, dtype=np.int64)
            blocks = np.concatenate([np.stack([cams, cams], 1),
                                     np.stack([t["blk_c1"].astype(np.int64), t["blk_c2"].astype(np.int64)], 1)], 0)
            if lin.packed.cc_costs:   # + the blocks the camera-camera costs put into the reduced system
                st = lin.packed.cc_structure
                ei, ej = st.edge_i.astype(np.int64), st.edge_j.astype(np.int64)
                blocks = np.concatenate([blocks, np.stack([np.maximum(ei, ej), np.minimum(ei, ej)], 1)], 0)
            self.pattern = TilePattern(nc, blocks, 6)
            self.sparse = self._want_sparse and self.pattern.l_tiles < self.pattern.ntiles * (self.pattern.ntiles + 1) // 2

    @property
    def info(self):
        """(B,) int32: non-zero where the damped system of a problem is not positive definite."""
        return self.info_chol + self.info_pts

    def check_info(self):
        bad = self.info.nonzero()
        if bad.numel():
            b = int(bad[0])
            raise RuntimeError(f"linalg.cholesky: (Batch element {b}): The factorization could not be completed because the "
                               "input is not positive-definite.")

    def _solve(self, damping, ellipsoidal_damping, damping_eps, check_info) -> torch.Tensor:
        lin = self.linearization
        if lin.g is None:
            raise RuntimeError("linearize() must be called before solve().")
        if damping is not None and isinstance(damping, torch.Tensor) and damping.ndim > 1:
            raise ValueError("Damping must be a float or a 1-D tensor.")
        self._ensure_buffers()
        lam = None
        if damping is not None:
            lam = self._lam
            if isinstance(damping, torch.Tensor):
                lam.copy_(damping.to(lam.dtype).expand(lam.shape[0]))
            else:
                lam.fill_(float(damping))
        p = lin.packed
        self.factor_version += 1
        self._factor_args = (lam.clone() if lam is not None else None, ellipsoidal_damping, damping_eps)
        if self.levels:
            K, t = self.K, self._level_t
            K.ba_schur_blocks(p.dstruct, lin.Hcc, lin.Hpp, lin.W, lin.gd, lam, ellipsoidal_damping, damping_eps, self.Sc,
                              t["diag_blk"], t["blk_dst"], self.rhs, self.Hinv, self.tvec, self.info_pts)
            K.vec_gather(self.rhs, self._xp, t["col_of_pad"])
            K.chol_factor_levels(self._level_layout, self.Sc, None, False, damping_eps, self.L, self.panels, self.info_chol,
                                 self.pattern, rhs=self._xp, y=self._yp)
            K.chol_solve_levels(self.L, self.panels, self._yp, self._xp, self.pattern, which=1)
            K.vec_gather(self._xp, self._dc, t["pad_of_col"])
            self.delta[:, :p.nc].copy_(self._dc)
            K.ba_backsub(p.dstruct, lin.W, self.Hinv, self.tvec, self.delta)
            if check_info:
                self.check_info()
            return self.delta
        self.K.ba_schur(p.dstruct, lin.Hcc, lin.Hpp, lin.W, lin.gd, lam, ellipsoidal_damping, damping_eps, self.S, self.rhs,
                        self.Hinv, self.tvec, self.info_pts)
        if p.cc_costs:
            # thx_ba_schur writes (does not accumulate) the blocks two cameras share through a point; the off-diagonal blocks of
            # the camera-camera costs are scatter-added on top.  Blocks only they touch would keep the previous call's sum: zeroed
            # first.
            if self._odo_only is None:
                self._odo_only = self._blocks_only_odometry(p) or ()
            if self._odo_only:
                self.S[:, self._odo_only[0], self._odo_only[1]] = 0
            bidx = torch.arange(self.S.shape[0], device=self.S.device).view(-1, 1, 1, 1)
            self.S.index_put_((bidx, lin._cc_rows.unsqueeze(0), lin._cc_cols.unsqueeze(0)), lin._cc_off, accumulate=True)
        if self.sparse:
            self.K.chol_factor_sparse(self.S, p.nc, None, False, damping_eps, self.L, self.panels, self.info_chol, self.pattern,
                                      rhs=self.rhs, y=self._y)
        else:
            self.K.chol_factor(self.S, p.nc, None, False, damping_eps, self.L, self.panels, self.info_chol, rhs=self.rhs,
                               y=self._y)
        if self.sparse:
            self.K.chol_solve_sparse(self.L, p.nc, self.panels, self._y, self._dc, self.pattern, backward_only=True)
        else:
            self.K.chol_solve_backward(self.L, p.nc, self.panels, self._y, self._dc)
        self.delta[:, :p.nc].copy_(self._dc)                                          # delta = [delta_c | delta_p]
        self.K.ba_backsub(p.dstruct, lin.W, self.Hinv, self.tvec, self.delta)
        if check_info:
            self.check_info()
        return self.delta


    def solve_with_factor(self, rhs: torch.Tensor) -> torch.Tensor:
        """(H + damping)^-1 rhs for another right-hand side (B, n) = [r_c | r_p] with the CACHED factor of the reduced camera
        system (the backward linear solve of the implicit mode): point elimination of rhs (thx_ba_schur re-run with rhs in
        place of g -- it rebuilds the same S, which is not factorised again), the two triangular solves with the cached L,
        back substitution."""
        lin = self.linearization
        p = lin.packed
        lam, ell, eps = self._factor_args
        gd = rhs.to(torch.float64).contiguous()
        rc = torch.empty_like(self.rhs)
        tv = torch.empty_like(self.tvec)
        scratch_info = torch.zeros_like(self.info_pts)
        if self.levels:
            K, t = self.K, self._level_t
            K.ba_schur_blocks(p.dstruct, lin.Hcc, lin.Hpp, lin.W, gd, lam, ell, eps, self.Sc, t["diag_blk"], t["blk_dst"], rc,
                              self.Hinv, tv, scratch_info)
            K.vec_gather(rc, self._xp, t["col_of_pad"])
            K.chol_solve_levels(self.L, self.panels, self._xp, self._xp, self.pattern, which=0)
            dc = torch.empty_like(rc)
            K.vec_gather(self._xp, dc, t["pad_of_col"])
            out = torch.empty(rhs.shape[0], lin.n, dtype=self.Sc.dtype, device=rhs.device)
            out[:, :p.nc].copy_(dc)
            K.ba_backsub(p.dstruct, lin.W, self.Hinv, tv, out)
            return out
        self.K.ba_schur(p.dstruct, lin.Hcc, lin.Hpp, lin.W, gd, lam, ell, eps, self.S, rc, self.Hinv, tv, scratch_info)
        dc = torch.empty_like(rc)
        if self.sparse:
            self.K.chol_solve_sparse(self.L, p.nc, self.panels, rc, dc, self.pattern)
        else:
            self.K.chol_solve(self.L, p.nc, self.panels, rc, dc)
        out = torch.empty(rhs.shape[0], lin.n, dtype=self.S.dtype, device=rhs.device)
        out[:, :p.nc].copy_(dc)
        self.K.ba_backsub(p.dstruct, lin.W, self.Hinv, tv, out)
        return out


class BAImplicitStep(torch.autograd.Function):
    """The grad-enabled last step of backward_mode="implicit" on a bundle-adjustment objective: one undamped Gauss-Newton step
    with the block Hessian built outside autograd (dense_linearization.py:61); backward = thx_se3_retract_vjp (cameras) /
    identity (points) -> one solve with the cached Schur factor -> thx_ba_vjp."""

    NAMES = ("feat", "w_obs", "focal", "k1", "k2", "log_radius", "cam_prior_target", "w_cam_prior", "pt_prior_target",
             "w_pt_prior")

    @staticmethod
    def forward(ctx, opt, packed, step, kwargs, *aux):
        import warnings
        aux, cc_aux = aux[:len(BAImplicitStep.NAMES)], aux[len(BAImplicitStep.NAMES):]   # (+ cc_meas, w_cc with odometry costs)
        solver = opt.linear_solver
        lin = solver.linearization
        lin._assemble()
        delta = solver._solve(None, False, 1e-8, check_info=False)   # plain GN; damped step of the optimizer if it fails
        if bool(solver.info.ne(0).any()):
            if kwargs.get("__strict_implicit_final_gn__", False):
                solver.check_info()
            warnings.warn("implicit backward: the undamped Gauss-Newton system is not positive definite, "
                          "falling back to the optimizer's damped step", RuntimeWarning)
            delta = opt.compute_delta(**kwargs)
            solver.check_info()
        delta = delta.clone()
        cams, pts = packed.tensors.cams.detach(), packed.tensors.points.detach()
        new = (torch.empty_like(cams), torch.empty_like(pts))
        packed.retract(delta, step, None, new)
        ctx.opt, ctx.packed, ctx.step, ctx.factor_version = opt, packed, step, solver.factor_version
        ctx.tensors = detached_ba_tensors(packed.tensors, cams, pts, aux)
        ctx.cc_tensors = None
        if cc_aux:
            ctx.cc_tensors = dataclasses.replace(packed.cc_tensors, poses=cams, meas=cc_aux[0].detach(), w_between=cc_aux[1].detach())
        ctx.delta = delta
        ctx.mark_non_differentiable(delta)
        return new[0], new[1], delta

    @staticmethod
    def backward(ctx, g_cams, g_pts, _g_delta):
        packed, solver = ctx.packed, ctx.opt.linear_solver
        lin, K, t = solver.linearization, solver.K, ctx.tensors
        if solver.factor_version != ctx.factor_version:
            raise RuntimeError("implicit backward: the cached factor of this forward pass was overwritten by a later "
                               "factorisation on the same optimizer; call backward() before the next forward().")
        B, n, nc = t.cams.shape[1], lin.n, packed.nc
        dt, dev = t.cams.dtype, t.cams.device
        gd = torch.zeros(B, n, dtype=dt, device=dev)
        if g_cams is not None:
            K.se3_retract_vjp(t.cams, ctx.delta, ctx.step, g_cams.contiguous(), gd)        # camera columns [0, 6C)
        if g_pts is not None:
            gd[:, nc:] = g_pts.permute(1, 0, 2).reshape(B, -1) * ctx.step                   # X + step * delta
        w = solver.solve_with_factor(gd)   # the backward linear solve
        grads = ba_vjp_grads(K, packed, t, w)
        if ctx.cc_tensors is not None:   # camera-camera Between costs: thx_pg_vjp over the camera columns of w
            ct, E = ctx.cc_tensors, len(packed.cc_costs)
            new = lambda *sh: torch.empty(*sh, dtype=dt, device=dev)  # noqa: E731
            g_meas, g_wb = new(E, B, 3, 4), new(E, B, 6)
            K.pg_vjp(packed.cc_dstruct, ct, w[:, :nc].contiguous(), g_meas, g_wb, new(1, B, 3, 4), new(1, B, 6))
            fit = lambda g_, like: g_.sum(1, keepdim=True) if like.shape[1] == 1 and B != 1 else g_  # noqa: E731
            grads = grads + (fit(g_meas, ct.meas), fit(g_wb, ct.w_between))
        return (None, None, None, None) + grads


def ba_unroll_backward(packed, solver, t, cc_t, factor_args, delta, grad_delta):
    """The part of an unrolled iteration's backward that both loops share (BAUnrolledIteration below, theseus_amd/plugin.py for
    the reference's loop): given grad_delta (B, n), rebuild this iteration's damped Schur system at the saved tensors ``t`` (the
    buffers were overwritten by the later iterations; same kernels, same bits), w = (H + D)^-1 grad_delta, thx_ba_unroll_vjp
    [+ thx_pg_unroll_vjp over the camera columns for camera-camera Between costs].  Returns (grad_cams (C,B,3,4), grad_points
    (Np,B,3), gradients of the auxiliary tensors in BAImplicitStep.NAMES order [+ cc_meas, w_cc])."""
    lin, K = solver.linearization, packed.K
    s = packed.structure
    C, B, nc = s.num_cams, t.cams.shape[1], packed.nc
    dt, dev = t.cams.dtype, t.cams.device
    lam, ell, eps = factor_args
    with packed.pinned(t, cc_t):
        HipSchurLinearizationCore._assemble(lin)
        solver._solve(lam, ell, eps, check_info=False)
        w = solver.solve_with_factor(grad_delta.contiguous())
    O, Kc, Kp = s.num_obs, s.num_cam_priors, s.num_pt_priors
    new = lambda *sh: torch.zeros(*sh, dtype=dt, device=dev)  # noqa: E731
    g = dict(cam_obs=new(max(O, 1), B, 3, 4), pt_obs=new(max(O, 1), B, 3), feat=new(max(O, 1), B, 2), w_obs=new(max(O, 1), B, 2),
             focal=new(max(O, 1), B), k1=new(max(O, 1), B), k2=new(max(O, 1), B),
             log_radius_obs=new(max(O, 1), B, 1) if t.robust_obs else None,
             cam_prior_cam=new(max(Kc, 1), B, 3, 4), cam_prior_target=new(max(Kc, 1), B, 3, 4), w_cam_prior=new(max(Kc, 1), B, 6),
             pt_prior_pt=new(max(Kp, 1), B, 3), pt_prior_target=new(max(Kp, 1), B, 3), w_pt_prior=new(max(Kp, 1), B, 3))
    ell_lam = lam if (lam is not None and ell) else None
    K.ba_unroll_vjp(packed.dstruct, t, w.contiguous(), delta, g, ell_damping=ell_lam)
    idx = lambda key, cnt: torch.from_numpy(np.asarray(s.t[key][:cnt], dtype=np.int64)).to(dev)  # noqa: E731
    oc, op = idx("obs_cam", O), idx("obs_pt", O)
    GC = torch.zeros_like(t.cams).index_add_(0, oc, g["cam_obs"][:O]).index_add_(0, idx("cam_prior_cam", Kc), g["cam_prior_cam"][:Kc])
    GP = torch.zeros_like(t.points).index_add_(0, op, g["pt_obs"][:O]).index_add_(0, idx("pt_prior_pt", Kp), g["pt_prior_pt"][:Kp])
    for k in ("focal", "k1", "k2"):      # calibration: per observation -> per camera
        g[k] = torch.zeros(C, B, dtype=dt, device=dev).index_add_(0, oc, g[k][:O]).unsqueeze(2)

    def fit(grad, count, like):   # (count, B, ...) -> the packed input's shape (count, 1|B, ...)
        if grad is None or like is None:
            return None
        grad = grad[:count]
        return grad.sum(1, keepdim=True) if like.shape[1] == 1 and B != 1 else grad
    grads = (fit(g["feat"], O, t.feat), fit(g["w_obs"], O, t.w_obs), fit(g["focal"], C, t.focal), fit(g["k1"], C, t.k1),
             fit(g["k2"], C, t.k2), fit(g["log_radius_obs"], O, t.log_radius_obs), fit(g["cam_prior_target"], Kc, t.cam_prior_target),
             fit(g["w_cam_prior"], Kc, t.w_cam_prior), fit(g["pt_prior_target"], Kp, t.pt_prior_target),
             fit(g["w_pt_prior"], Kp, t.w_pt_prior))
    if cc_t is not None:   # camera-camera Between costs: thx_pg_unroll_vjp over the camera columns of w, delta
        st = packed.cc_structure
        E = st.num_edges
        gpi, gpj, gm, gwb = new(E, B, 3, 4), new(E, B, 3, 4), new(E, B, 3, 4), new(E, B, 6)
        K.pg_unroll_vjp(packed.cc_dstruct, cc_t, w[:, :nc].contiguous(), delta[:, :nc].contiguous(), gpi, gpj, gm, gwb,
                        new(1, B, 3, 4), new(1, B, 3, 4), new(1, B, 6), ell_damping=ell_lam)
        ei = torch.from_numpy(st.edge_i.astype(np.int64)).to(dev)
        ej = torch.from_numpy(st.edge_j.astype(np.int64)).to(dev)
        GC = GC.index_add_(0, ei, gpi).index_add_(0, ej, gpj)
        grads = grads + (fit(gm, E, cc_t.meas), fit(gwb, E, cc_t.w_between))
    return GC, GP, grads


class BAUnrolledIteration(torch.autograd.Function):
    """One DIFFERENTIATED iteration of a bundle-adjustment objective (BackwardMode.UNROLL / TRUNCATED,
    theseus/optimizer/nonlinear/nonlinear_least_squares.py:223-292: the Hessian is part of the graph):
    ``cams_new = cams exp(step * delta_c)``, ``points_new = points + step * delta_p``, ``delta = (H + D)^-1 g`` by point elimination.
    Forward: the optimizer's own kernels at the detached iterate.  Backward, given the gradients of the new state: thx_se3_retract_vjp /
    identity -> grad_delta; Compose.backward's matrix rule / identity -> the direct path; the iteration's Schur system is REBUILT at
    the saved iterate with the saved damping (same kernels, same bits: the factor of a (B, 6C, 6C) reduced system per iteration is
    not kept) and solved for w = (H + D)^-1 grad_delta; thx_ba_unroll_vjp(w, delta) [+ thx_pg_unroll_vjp over the camera columns for
    camera-camera Between costs] -> per-cost gradients, summed per camera / point."""

    @staticmethod
    def forward(ctx, opt, packed, frozen, kwargs, cams, pts, *aux):
        aux, cc_aux = aux[:len(BAImplicitStep.NAMES)], aux[len(BAImplicitStep.NAMES):]
        solver = opt.linear_solver
        lin, K = solver.linearization, packed.K
        cd, pd = cams.detach().contiguous(), pts.detach().contiguous()
        t = packed.tensors
        t.cams, t.points = cd, pd                       # the kernels linearise at the tail loop's iterate
        lin._assemble()
        delta = opt.compute_delta(**kwargs)
        if bool(solver.info.ne(0).any()):
            try:
                solver.check_info()
            except RuntimeError as run_err:
                raise RuntimeError(f"There was an error while running the linear optimizer. Original error message: {run_err}. "
                                   "Backward pass will not work. To obtain the best solution seen before the error, run with "
                                   "torch.no_grad()") from None
        step = float(opt.params.step_size)
        mask = frozen.to(torch.uint8).contiguous() if frozen is not None else None
        new = (torch.empty_like(cd), torch.empty_like(pd))
        packed.retract(delta, step, mask, new)
        ctx.opt, ctx.packed, ctx.step, ctx.frozen = opt, packed, step, frozen
        ctx.factor_args = solver._factor_args            # (lambda clone | None, ellipsoidal, eps) of THIS iteration
        ctx.tensors = detached_ba_tensors(t, cd, pd, aux)
        ctx.cc_tensors = None
        if cc_aux:
            ctx.cc_tensors = dataclasses.replace(packed.cc_tensors, poses=cd, meas=cc_aux[0].detach(), w_between=cc_aux[1].detach())
        ctx.delta = delta.detach().clone()
        ctx.mark_non_differentiable(delta)
        return new[0], new[1], delta

    @staticmethod
    def backward(ctx, g_cams, g_pts, _g_delta):
        packed, solver = ctx.packed, ctx.opt.linear_solver
        lin, K, t, step, delta = solver.linearization, packed.K, ctx.tensors, ctx.step, ctx.delta
        s = packed.structure
        C, Np, B, n, nc = s.num_cams, s.num_points, t.cams.shape[1], lin.n, packed.nc
        dt, dev = t.cams.dtype, t.cams.device
        g_cams = torch.zeros_like(t.cams) if g_cams is None else g_cams.contiguous()
        g_pts = torch.zeros_like(t.points) if g_pts is None else g_pts.contiguous()
        gd = torch.zeros(B, n, dtype=dt, device=dev)
        K.se3_retract_vjp(t.cams, delta, step, g_cams, gd)                                # camera columns [0, 6C)
        gd[:, nc:] = g_pts.permute(1, 0, 2).reshape(B, -1) * step                         # X + step * delta
        from .autograd import compose_left_backward
        GC = compose_left_backward(K, "SE3", t.cams, delta[:, :nc].contiguous(), step, g_cams)
        GP = g_pts
        if ctx.frozen is not None:           # frozen problems: state_new = state
            fz = ctx.frozen.bool()
            gd = gd * (~fz).to(dt).view(-1, 1)
            GC = torch.where(fz.view(1, B, 1, 1), g_cams, GC)
        gC, gP, grads = ba_unroll_backward(packed, solver, t, ctx.cc_tensors, ctx.factor_args, delta, gd)
        GC, GP = GC + gC, GP + gP
        return (None, None, None, None, GC, GP) + grads


def ba_vjp_grads(K, packed, t, w):
    """Gradients of phi = w^T g(theta) (g = A^T b of the bundle-adjustment linearization at the tensors ``t``, Hessian
    detached) w.r.t. the packed auxiliary tensors, in ``BAImplicitStep.NAMES`` order and in the shapes they were packed in:
    thx_ba_vjp + the per-observation -> per-camera reduction of the calibration gradients."""
    s = packed.structure
    B = t.cams.shape[1]
    dt, dev = t.cams.dtype, t.cams.device
    O, Kc, Kp = s.num_obs, s.num_cam_priors, s.num_pt_priors
    new = lambda *sh: torch.empty(*sh, dtype=dt, device=dev)  # noqa: E731
    g = dict(feat=new(max(O, 1), B, 2), w_obs=new(max(O, 1), B, 2), focal=new(max(O, 1), B), k1=new(max(O, 1), B),
             k2=new(max(O, 1), B), log_radius=new(max(O, 1), B, 1) if t.robust_obs else None,
             cam_prior_target=new(max(Kc, 1), B, 3, 4), w_cam_prior=new(max(Kc, 1), B, 6),
             pt_prior_target=new(max(Kp, 1), B, 3), w_pt_prior=new(max(Kp, 1), B, 3))
    K.ba_vjp(packed.dstruct, t, w.contiguous(), g)
    # calibration: per observation -> per camera
    oc = torch.from_numpy(np.asarray(s.t["obs_cam"][:O], dtype=np.int64)).to(dev)
    C = s.num_cams
    for k in ("focal", "k1", "k2"):
        g[k] = torch.zeros(C, B, dtype=dt, device=dev).index_add_(0, oc, g[k][:O]).unsqueeze(2)

    def fit(grad, count, like):   # (count, B, ...) -> the packed input's shape (count, 1|B, ...)
        if grad is None or like is None:
            return None
        grad = grad[:count]
        return grad.sum(1, keepdim=True) if like.shape[1] == 1 and B != 1 else grad
    counts = dict(feat=O, w_obs=O, focal=C, k1=C, k2=C, log_radius=O, cam_prior_target=Kc, w_cam_prior=Kc,
                  pt_prior_target=Kp, w_pt_prior=Kp)
    likes = dict(feat=t.feat, w_obs=t.w_obs, focal=t.focal, k1=t.k1, k2=t.k2, log_radius=t.log_radius_obs,
                 cam_prior_target=t.cam_prior_target, w_cam_prior=t.w_cam_prior, pt_prior_target=t.pt_prior_target,
                 w_pt_prior=t.w_pt_prior)
    return tuple(fit(g[k], counts[k], likes[k]) for k in BAImplicitStep.NAMES)


def detached_ba_tensors(tensors, cams, points, aux):
    """The packed tensors with cams / points replaced and the auxiliary tensors (BAImplicitStep.NAMES order) detached."""
    keys = ("feat", "w_obs", "focal", "k1", "k2", "log_radius_obs", "cam_prior_target", "w_cam_prior", "pt_prior_target",
            "w_pt_prior")
    return dataclasses.replace(tensors, cams=cams, points=points,
                               **{k: (None if a is None else a.detach()) for k, a in zip(keys, aux)})


def ba_implicit_step(opt, packed, step: float, kwargs):
    """-> ((cams_new, points_new), delta): called by NonlinearLeastSquares._implicit_last_step under the caller's grad mode."""
    packed.flush_variables()
    packed.sync(force=True)   # re-pack the auxiliary tensors WITH their autograd history
    t = packed.tensors
    cc = (packed.cc_tensors.meas, packed.cc_tensors.w_between) if packed.cc_costs else ()
    cams, pts, delta = BAImplicitStep.apply(opt, packed, step, kwargs, t.feat, t.w_obs, t.focal, t.k1, t.k2, t.log_radius_obs,
                                            t.cam_prior_target, t.w_cam_prior, t.pt_prior_target, t.w_pt_prior, *cc)
    return (cams, pts), delta


class HipSchurSolver(HipSchurSolverCore, LinearSolver):
    def __init__(self, objective: Objective, linearization_cls: Optional[Type[Linearization]] = None,
                 linearization_kwargs: Optional[Dict[str, Any]] = None, sparse_reduced_system: bool = True,
                 ordering: str = "auto", **kwargs):
        """``sparse_reduced_system=False``: factorise the reduced camera system as a dense matrix even where its tile pattern
        has holes (for comparisons; the result is the same bit for bit).  ``ordering``: "auto" | "nd" | "md" | "natural" -- the
        camera order of the reduced system's factorisation (HipSchurSolverCore._schur_solver_init)."""
        linearization_cls = linearization_cls or HipSchurLinearization
        if not (isinstance(linearization_cls, type) and issubclass(linearization_cls, HipSchurLinearization)):
            raise RuntimeError(f"HipSchurSolver only works with HipSchurLinearization, but {linearization_cls} was provided.")
        LinearSolver.__init__(self, objective, linearization_cls, linearization_kwargs)
        self._schur_solver_init(sparse_reduced_system, ordering)

    def solve(self, damping: Optional[Union[float, torch.Tensor]] = None, ellipsoidal_damping: bool = True,
              damping_eps: float = 1e-8, check_info: bool = True, **kwargs) -> torch.Tensor:
        return self._solve(damping, ellipsoidal_damping, damping_eps, check_info)
