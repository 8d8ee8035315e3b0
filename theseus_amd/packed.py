"""Packed device representation of a pose-graph objective (all poses SE3, or all poses SE2).

This is the analogue of the reference's ``Vectorize`` pass (theseus/core/vectorizer.py:112-474): it
groups the objective's cost functions by schema (Between / Difference), but instead of
stacking tensors for batched ATen calls it keeps ONE entity-major buffer per role
(poses (P,B,3,4 | 4), measurements (E,Bm,3,4 | 4), weights (E,Bw,6 | 3), prior targets, prior weights) that
the fused HIP kernels index directly.  Variables are tracked by ``_num_updates`` so that the
buffers are re-packed only when somebody changed a variable behind our back.
"""
import weakref
from typing import Optional

import numpy as np
import torch

from .compiler import PoseGraphStructure
from .core import Objective, Variable
from .kernels import PGTensors, default_kernels, fast_approx_local_jacobians, round_up, _lib

ERR_CHUNKS = _lib.THX_ERR_CHUNKS


import operator

_GET_TENSOR = {True: operator.attrgetter("_tensor"), False: operator.attrgetter("tensor")}   # own Variable: skip the property
_DATA_PTR, _VERSION, _NUM_UPDATES = operator.methodcaller("data_ptr"), operator.attrgetter("_version"), operator.attrgetter("_num_updates")

# one C++ loop over a list of tensors instead of 33 k method calls (torch's cudagraph trees check their static inputs with it)
_PTRS_EQUAL = getattr(torch._C, "_tensors_data_ptrs_at_indices_equal", None)


class _AuxDeepStamp:
    """Deep stamp of the AUXILIARY variables' tensors (measurements, weights, ... -- 33 k of them in a bundle-adjustment objective):
    has anybody replaced a tensor object, swapped the storage under one (``var.tensor.data = other`` / ``set_()``: object and
    version counter stay), or edited one in place (``var.tensor.mul_()``: only the autograd version counter moves)?  The reference
    re-reads ``var.tensor`` at every evaluation (core/objective.py:813-830) and sees all three.

    Three flat comparisons instead of a 3-tuple per tensor (12 ms at 33 k tensors):
      * the tuple of object ids (0.9 ms; the tensors are kept referenced here, so an id cannot be recycled);
      * the storage pointers, in one C++ call against the cached list (2 ms; a Python pass of ``data_ptr()`` calls: 4.3 ms);
      * the version counters of one REPRESENTATIVE per group of tensors that share one: views share their base's counter
        (``feat[:, k]`` for 32768 observations is one counter), so an in-place edit through ANY of them moves the representative's.
        A tensor's ``_base`` is fixed at creation: the grouping holds for as long as the ids do."""

    def __init__(self):
        self.refs = self.ids = self.ptrs = self.ptrs_t = self.reps = self.idx = None

    def stamp(self, ts):
        ids = tuple(map(id, ts))
        if ids != self.ids:
            self.refs, self.ids = ts, ids
            self.ptrs = list(map(_DATA_PTR, ts))
            self.ptrs_t = tuple(self.ptrs)
            self.idx = list(range(len(ts)))
            groups = {}
            for t in ts:
                base = t._base
                groups.setdefault(id(base) if base is not None else id(t), t)
            self.reps = list(groups.values())
        else:
            same = None
            if _PTRS_EQUAL is not None and ts:
                try:
                    same = _PTRS_EQUAL(ts, self.ptrs, self.idx)
                except TypeError:      # (a private torch symbol: another signature in another release -> the Python pass)
                    same = None
            if same is None:
                same = list(map(_DATA_PTR, ts)) == self.ptrs
            if not same:
                self.ptrs = list(map(_DATA_PTR, ts))
                self.ptrs_t = tuple(self.ptrs)
        return ids, self.ptrs_t, tuple(map(_VERSION, self.reps))


def _opt_deep_stamp(ts):
    """Deep stamp of the optimisation variables' tensors: storage pointer + version counter (they are views of the packed state:
    holding the objects would pin a state buffer), as two flat tuples."""
    return tuple(map(_DATA_PTR, ts)), tuple(map(_VERSION, ts))


def _views_deep_stamp(buffers):
    """``_opt_deep_stamp`` of variables that were JUST re-pointed at ``buf.unbind(0)`` of these buffers, without touching the 8.7 k
    view objects: view k starts ``k * stride(0)`` elements into its buffer and shares the buffer's version counter."""
    ptrs, vers = (), ()
    for buf in buffers:
        n, step = buf.shape[0], buf.stride(0) * buf.element_size()
        base = buf.data_ptr()
        ptrs += tuple(range(base, base + n * step, step)) if step else (base,) * n
        vers += (buf._version,) * n
    return ptrs, vers


class UnsupportedObjective(NotImplementedError):
    pass


# The packer recognises objects by the reference's CLASS NAMES, so that it accepts theseus_amd's mirror classes
# and the real ``theseus`` ones alike (theseus_amd/plugin.py plugs this back end into the reference's own loop).
def _kind(obj) -> str:
    names = {c.__name__ for c in type(obj).__mro__}
    for k in ("SE3", "SE2", "SO3", "SO2", "Between"):
        if k in names:
            return k
    if "Local" in names or "Difference" in names:
        return "Difference"
    return type(obj).__name__


GROUP_SHAPE = {"SE3": (3, 4), "SE2": (4,), "SO3": (3, 3), "SO2": (2,)}
GROUP_DOF = {"SE3": 6, "SE2": 3, "SO3": 3, "SO2": 1}


def _weight_diag(w, dof: int) -> torch.Tensor:
    """(Bw, dof) sqrt-information diagonal of a Scale/DiagonalCostWeight (theseus/core/cost_weight.py:60-139)."""
    if hasattr(w, "sqrt_diag"):
        return w.sqrt_diag(dof)
    names = {c.__name__ for c in type(w).__mro__}
    if "ScaleCostWeight" in names:
        return w.scale.tensor.view(-1, 1).expand(-1, dof)
    if "DiagonalCostWeight" in names:
        d = w.diagonal.tensor
        if d.shape[1] != dof:
            raise ValueError(f"This cost needs a {dof}-dimensional DiagonalCostWeight.")
        return d
    raise UnsupportedObjective(f"HIP backend supports Scale/DiagonalCostWeight; got {type(w).__name__}. "
                               "There is no CPU/eager fallback.")


_LOSS_KIND = {"WelschLoss": _lib.LOSS_WELSCH, "HuberLoss": _lib.LOSS_HUBER, "HingeLoss": _lib.LOSS_HINGE,
              "GemanMcClureLoss": _lib.LOSS_GEMAN_MCCLURE}


class _GNCRadius:
    """The radius entry of a GNCRobustCostFunction (robust_cost_function.py:173-222) as the kernels take it: log(mu * radius) =
    log_loss_radius + log(gnc_control_val) -- Geman-McClure depends on the two only through their product (robust_loss.py:96-113).
    Built with torch ops at packing time, so gradients reach both variables; ``vars`` are what the packers track for edits."""

    def __init__(self, log_radius, mu):
        self.vars = (log_radius, mu)

    @property
    def tensor(self):
        lr, mu = self.vars[0].tensor, self.vars[1].tensor
        return lr.reshape(lr.shape[0], -1)[:, :1] + mu.reshape(mu.shape[0], -1)[:, :1].log()


def _radius_vars(r):
    """the Variables behind a radius entry (None | Variable | _GNCRadius)"""
    return () if r is None else (r.vars if isinstance(r, _GNCRadius) else (r,))


def _unwrap_robust(c):
    """cost -> (base cost, loss code, log_loss_radius variable [_GNCRadius for a GNC cost] | None); code = _lib.LOSS_* |
    _lib.LOSS_FLATTEN for ``flatten_dims=True``  (theseus/core/robust_cost_function.py:52-85, 173-222)."""
    if "RobustCostFunction" not in {k.__name__ for k in type(c).__mro__}:
        return c, _lib.LOSS_NONE, None
    kind = _LOSS_KIND.get(type(c.loss).__name__)
    if kind is None:
        raise UnsupportedObjective(f"HIP backend fuses WelschLoss / HuberLoss / HingeLoss / GemanMcClureLoss; got "
                                   f"{type(c.loss).__name__} ({c.name}).  There is no CPU/eager fallback.")
    radius = c.log_loss_radius
    if kind == _lib.LOSS_GEMAN_MCCLURE:
        radius = _GNCRadius(c.log_loss_radius, c.gnc_control_val)
    return c.cost_function, kind | (_lib.LOSS_FLATTEN if c.flatten_dims else 0), radius


def _role_codes(codes):
    """loss codes of one cost role -> (role code, per-cost table | None): one code when all costs agree, else the first non-zero
    code + the (count,) int32 table (include/theseus_hip.h: thx_pg_data.loss_between / loss_prior)."""
    if len(set(codes)) <= 1:
        return (codes[0] if codes else _lib.LOSS_NONE), None
    return next(c for c in codes if c), np.asarray(codes, dtype=np.int32)


def _aux_vars(x):
    a = x.aux_vars
    return list(a() if callable(a) else a)


class PackedPoseGraph:
    def __init__(self, objective: Objective, kernels=None, order=None):
        """``order``: names of the optimisation variables in COLUMN order (a VariableOrdering -- e.g. the fill-reducing one of
        theseus_amd/sparse.py); default = insertion order (theseus/optimizer/variable_ordering.py:19-27)."""
        self.objective = objective
        self.K = kernels or default_kernels()
        self.pose_vars = []
        self.group = None
        self.order = tuple(order) if order is not None else tuple(objective.optim_vars.keys())
        if sorted(self.order) != sorted(objective.optim_vars.keys()):
            raise ValueError("the variable ordering must hold every optimisation variable of the objective exactly once")
        for v in (objective.optim_vars[name] for name in self.order):
            kind = _kind(v)
            if kind not in GROUP_SHAPE or (self.group is not None and kind != self.group):
                raise UnsupportedObjective(
                    f"HIP backend fuses objectives whose optimisation variables are all SE3, all SE2, all SO3 or all SO2; got "
                    f"{type(v).__name__} ({v.name}). There is no CPU/eager fallback.")
            self.group = kind
            self.pose_vars.append(v)
        self.gshape = GROUP_SHAPE[self.group]
        self.dof = GROUP_DOF[self.group]
        index = {v.name: k for k, v in enumerate(self.pose_vars)}
        edges, priors, e_rows, p_rows = [], [], [], []
        self.edge_costs = []
        self.prior_costs = []
        row = 0
        self.edge_radius, self.prior_radius = [], []   # log_loss_radius Variable per robust cost
        codes = {"Between": [], "Difference": []}
        for wrapped in objective.cost_functions.values():
            c, loss, radius = _unwrap_robust(wrapped)
            if _kind(c) == "Between":
                edges.append((index[c.v0.name], index[c.v1.name]))
                e_rows.append(row)
                self.edge_costs.append(c)
                self.edge_radius.append(radius)
                codes["Between"].append(loss)
            elif _kind(c) == "Difference":
                priors.append(index[c.var.name])
                p_rows.append(row)
                self.prior_costs.append(c)
                self.prior_radius.append(radius)
                codes["Difference"].append(loss)
            else:
                raise UnsupportedObjective(
                    f"HIP backend has no fused kernel for cost function {type(c).__name__} ({c.name}); "
                    "supported: Between, Difference/Local on SE3 / SE2 / SO3.  There is no CPU/eager fallback.")
            row += c.dim()
        self._refuse_fast_approx(UnsupportedObjective)
        # plain, Welsch, Huber and flatten_dims costs may be mixed inside one role: per-cost loss table
        self.robust_between, self.loss_between = _role_codes(codes["Between"])
        self.robust_prior, self.loss_prior = _role_codes(codes["Difference"])
        self.structure = PoseGraphStructure.build(len(self.pose_vars), edges, priors, e_rows, p_rows, dof=self.dof)
        self.n = self.structure.num_cols
        self.m = self.structure.num_rows
        self.ld = round_up(self.n, 32)
        self.version = objective.current_version
        self.tensors: Optional[PGTensors] = None
        # the O(1) "nothing changed" test relies on theseus_amd.core.Variable's global update counter
        self._own_variables = all(isinstance(v, Variable) for v in self.tracked_list())
        self._stamp = None
        self._deep_stamp = None
        self._aux_stamp = _AuxDeepStamp()
        self._keep_graph_tensors = False   # set by the optimizer around an optimize() that differentiates from the initial tensors
        self._defer_repoint = False        # set by the optimizer around its first sync (see sync)
        self._global_stamp = -1
        self._vars_stale = False
        self._state_exposed = False
        self._known = []
        self._scratch = {}

    def _refuse_fast_approx(self, exc=NotImplementedError):
        """``fast_approx_local_jacobians`` (theseus/embodied/misc/local_cost_fn.py:43-57: identity Jacobians for Local /
        Difference costs) is not fused into the kernels: refuse loudly instead of silently computing the exact Jacobian.
        (At construction the plugin for the real ``theseus`` catches this and takes its generic block path, where the
        reference evaluates its own -- approximate -- Jacobians.)"""
        if self.prior_costs and fast_approx_local_jacobians():
            raise exc("HIP backend: the global option fast_approx_local_jacobians=True is not fused into the pose-graph "
                      "kernels (Difference/Local costs would silently get their exact Jacobian).  There is no CPU/eager "
                      "fallback; unset it, or set it BEFORE constructing the optimizer when running the real theseus "
                      "through theseus_amd.plugin (generic block path).")

    # ---- packing ----------------------------------------------------------------------------
    def _tracked(self):
        for v in self.pose_vars:
            yield v
        for c in self.edge_costs:
            yield c.measurement
            yield from _aux_vars(c.weight)
        for c in self.prior_costs:
            yield c.target
            yield from _aux_vars(c.weight)
        for r in self.edge_radius + self.prior_radius:
            yield from _radius_vars(r)

    def tracked_list(self):
        """Every variable the packed buffers are built from, each ONCE (a cost weight shared by 1024 edges is one entry), the
        optimisation variables (poses) first.  The objective is frozen once an optimizer holds it (Optimizer.optimize checks its
        version): the walk over the cost functions is done once."""
        tracked = self.__dict__.get("_tracked_list")
        if tracked is None:
            seen, tracked = set(), []
            for v in self._tracked():
                if id(v) not in seen:
                    seen.add(id(v))
                    tracked.append(v)
            self._tracked_list = tracked
        return tracked

    def _counters_unchanged(self) -> bool:
        return self._own_variables and self._stamp is not None and Variable._global_updates == self._global_stamp

    def _current_stamp(self, deep: bool = False, count: Optional[int] = None):
        # the objective is frozen once an optimizer holds it (Optimizer.optimize checks its version): the walk over the cost
        # functions is done once, later stamps are one pass over the cached list (7 k variables at the headline size)
        tracked = self.tracked_list()
        if count is not None:
            tracked = tracked[:count]
        if deep:
            # (optimisation part, auxiliary part): catches IN-PLACE edits of a variable's tensor that never go through
            # Variable.update() (an nn.Parameter stepped by a torch optimizer, ``var.tensor.mul_()``), ``var.tensor = other`` and
            # ``var.tensor.data = other`` -- see _AuxDeepStamp.  ``count``: the optimisation part alone.
            ts = list(map(_GET_TENSOR[self._own_variables], tracked))
            n_opt = min(len(self.pose_vars), len(ts))
            opt = _opt_deep_stamp(ts[:n_opt])
            return opt if count is not None else (opt, self._aux_stamp.stamp(ts[n_opt:]))
        return tuple(map(_NUM_UPDATES, tracked))

    @staticmethod
    def _stack(ts, B):
        """list of (b_k, ...) with b_k in {1,B} -> (len, 1|B, ...) contiguous."""
        if not ts:
            return None
        if all(t.shape[0] == ts[0].shape[0] for t in ts):
            return torch.stack(ts, dim=0).contiguous()
        return torch.stack([t.expand(B, *t.shape[1:]) for t in ts], dim=0).contiguous()

    def sync(self, force: bool = False, deep: bool = False):
        """(Re)pack the variable tensors into the device buffers if any variable changed.  ``deep`` also looks for in-place
        edits of the variables' tensors (storage pointer + version counter): the optimizers ask for it once per
        ``optimize()``, the inner-loop calls keep the O(1) test; with the reference's own Variable class (theseus_amd/
        plugin.py) every call is deep -- there is no global update counter to lean on.  Poses and auxiliary tensors are
        re-packed independently (the reference's loop replaces the pose tensors every iteration, the measurements never)."""
        deep = deep or not self._own_variables
        if (not force and not deep and self.tensors is not None
                and Variable._global_updates == self._global_stamp):
            return  # nobody called Variable.update()/to() since the last look: O(1) fast path
        # (nobody called Variable.update() / to() since the last look: the update counters -- the shallow stamp -- are what they
        #  were; a pass over 42 k variables of a bundle-adjustment objective is ~10 ms of host time per optimize())
        # (the reference's Variable: the deep stamp -- storage + version of every tensor -- is the only one looked at: a
        #  Variable.update() that changes neither is not a change.  One pass per call instead of two.)
        shallow = self._own_variables or self._stamp is None
        stamp = self._stamp if (self._counters_unchanged() or not shallow) else self._current_stamp()
        dstamp = self._current_stamp(deep=True) if deep else None
        nP = len(self.pose_vars)
        if force or self.tensors is None:
            poses_changed = aux_changed = True
        else:
            poses_changed = (shallow and stamp[:nP] != self._stamp[:nP]) or (deep and dstamp[0] != self._deep_stamp[0])
            aux_changed = (shallow and stamp[nP:] != self._stamp[nP:]) or (deep and dstamp[1] != self._deep_stamp[1])
        if not (poses_changed or aux_changed):
            self._global_stamp = Variable._global_updates
            return
        self.flush_variables()
        obj = self.objective
        if not (self._own_variables and obj._batch_size_is_current()):   # (Objective.update just walked the variables)
            obj._resolve_batch_size()
        B = obj.batch_size
        dev, dt = self.pose_vars[0].device, obj.dtype
        gs, dof = self.gshape, self.dof
        adopted = False
        if poses_changed or self.tensors.poses.shape[1] != B:
            ts = [v.tensor for v in self.pose_vars]
            poses = self.buffer_of(ts)   # the variables already view ONE packed buffer (a kernel's output): no copy
            adopted = poses is not None
            if poses is None:
                poses = self._stack([t.expand(B, *gs) if t.shape[0] != B else t for t in ts], B)
        else:
            poses = self.tensors.poses
        if aux_changed or self.tensors.batch != B:
            E, Kp = self.structure.num_edges, self.structure.num_priors
            empty = lambda *s: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
            meas = self._stack([c.measurement.tensor for c in self.edge_costs], B) if E else empty(0, 1, *gs)
            wb = self._stack([_weight_diag(c.weight, dof) for c in self.edge_costs], B) if E else empty(0, 1, dof)
            tgt = self._stack([c.target.tensor for c in self.prior_costs], B) if Kp else empty(0, 1, *gs)
            wp = self._stack([_weight_diag(c.weight, dof) for c in self.prior_costs], B) if Kp else empty(0, 1, dof)
            unused = empty(1, 1)   # the log_radius slot of a plain cost inside a mixed role (never read)
            radii = lambda rs: self._stack([unused if r is None else r.tensor.view(-1, 1) for r in rs], B)  # noqa: E731
            table = lambda a: None if a is None else torch.from_numpy(a).to(dev)  # noqa: E731
            self.tensors = PGTensors(poses=poses, meas=meas, w_between=wb, prior_target=tgt, w_prior=wp,
                                     robust_between=self.robust_between,
                                     log_radius_between=radii(self.edge_radius) if self.robust_between else None,
                                     robust_prior=self.robust_prior,
                                     log_radius_prior=radii(self.prior_radius) if self.robust_prior else None,
                                     loss_between=table(self.loss_between), loss_prior=table(self.loss_prior))
        else:
            self.tensors.poses = poses
        # (the auxiliary entries of the deep stamp: what this call saw -- _repoint_variables patches the poses' entries)
        self._deep_stamp = dstamp if dstamp is not None else self._current_stamp(deep=True)
        if adopted:   # the variables hold these very views already: only the stamps move
            self._stamp = stamp
            self._global_stamp = Variable._global_updates
            self._vars_stale = False
            self._state_exposed = True
        elif (not poses.requires_grad and (self._keep_graph_tensors or not self._own_variables)
              and any(v.tensor.requires_grad for v in self.pose_vars)):
            # packed under no_grad from tensors that carry autograd history (the values the caller passed in, about to be
            # differentiated through: backward_mode="unroll"): re-pointing the variables to views of this graph-less copy would cut
            # them off -- they keep their tensors (same values), the copy is private.  Only where that is needed: theseus_amd's
            # own loop says so for the optimize() calls that captured the initial tensors (``_keep_graph_tensors``: everywhere
            # else the variables view the packed state, also inside callbacks and after an exception); under the reference's
            # loop (its Variable class), which re-assigns every variable every iteration anyway.
            self._stamp = stamp
            self._global_stamp = Variable._global_updates
            self._vars_stale = False
            self._state_exposed = False
        elif self._defer_repoint and self._own_variables:
            # the optimizer's sync at the start of optimize(): the loop is about to continue on a PRIVATE state buffer and re-points
            # the variables when it is done (or before anybody can look: callbacks, exceptions -- flush_variables) -- re-pointing
            # them to views of THIS buffer first was 5 ms of host time per optimize() at 4096 poses in front of the first kernel,
            # for views nobody ever read.  The variables keep the tensors they hold (the same values) until then.
            self._stamp = stamp
            self._global_stamp = Variable._global_updates
            self._vars_stale = True
            self._state_exposed = False
        else:
            self._repoint_variables()

    # ---- buffers whose per-pose views are known (so that a list of variable tensors can be recognised as one buffer) ----
    def remember_views(self, buf: torch.Tensor, views):
        # A WEAK reference to the buffer: remembering it must not keep it (50 MB at the headline size) alive -- the caching
        # allocator would have to hipMalloc fresh state buffers in every optimize() (measured: +20 ms per optimize).  The views
        # are remembered by object id + storage pointer, not by weak references: a weakref to a tensor costs ~4 us to create,
        # 35 ms per TheseusLayer.forward at 4096 poses (profiles/r6/).  A recycled id alone cannot pass for a view: the same
        # object would also have to start at the same address inside the (still alive) buffer.
        self._known = [(weakref.ref(buf), tuple(map(id, views)), list(map(_DATA_PTR, views)))] + self._known[:1]

    def buffer_of(self, tensors):
        ids = None
        for wbuf, vids, ptrs in self._known:
            buf = wbuf()
            if buf is None or len(vids) != len(tensors):
                continue
            if ids is None:
                ids = tuple(map(id, tensors))
            if ids == vids and list(map(_DATA_PTR, tensors)) == ptrs and tensors[0].shape == buf.shape[1:]:
                return buf
        return None

    def _repoint_variables(self):
        """Make every optimisation variable's tensor a view of the packed pose buffer."""
        poses = self.tensors.poses
        # the optimiser calls this under no_grad; after the implicit last step the pose buffer carries a graph
        # and the per-variable views must stay attached to it
        with torch.set_grad_enabled(poses.requires_grad):
            views = poses.unbind(0)  # one call builds all the views
            attr = "_tensor" if self._own_variables else "tensor"   # (own Variable: no update count for a re-pointing, core.py)
            for v, t in zip(self.pose_vars, views):
                setattr(v, attr, t)
        self.remember_views(poses, views)
        if not self._counters_unchanged():
            self._stamp = self._current_stamp()
        # deep stamp: only the optimisation variables were re-pointed here -- the auxiliary variables' entries (the bulk: 1 k
        # measurements of a pose graph, 33 k features of a bundle-adjustment problem) are still the ones sync() looked at
        n_opt = len(self.pose_vars)
        old = self._deep_stamp
        if old is not None:
            self._deep_stamp = (_views_deep_stamp((poses,)), old[1])
        else:
            self._deep_stamp = self._current_stamp(deep=True)
        self._global_stamp = Variable._global_updates
        self._vars_stale = False
        self._state_exposed = True   # the variables (and whoever holds their tensors) now view the state buffer

    def privatize_state(self):
        """Called at the start of ``optimize()``: if the state buffer is the one the variables' tensors view (i.e. what the
        previous ``optimize()`` / ``forward()`` handed to the user, possibly carrying an autograd graph), continue on a
        private copy -- the loop recycles its two state buffers through raw-pointer kernels and must never write into a
        tensor somebody else holds."""
        if self._state_exposed:
            with torch.no_grad():
                self.tensors.poses = self.tensors.poses.detach().clone()
            self._state_exposed = False
            self._vars_stale = True

    def set_poses(self, poses: torch.Tensor, repoint: bool = True):
        """Adopt a new packed pose buffer (after an accepted LM step).  ``repoint=False`` defers the
        O(#variables) Python re-pointing of the Variable objects (the optimiser loop flushes before anybody
        can look: callbacks, loop exit)."""
        self.tensors.poses = poses
        if repoint:
            self._repoint_variables()
        else:
            self._vars_stale = True

    def flush_variables(self):
        if self._vars_stale and self.tensors is not None:
            self._repoint_variables()

    # ---- BackwardMode.UNROLL / TRUNCATED (theseus_amd/autograd.py:PGUnrolledIteration) ---------------------------------------
    def prepare_unroll(self):
        """Before the differentiable tail loop: re-pack the auxiliary tensors WITH their autograd history (once)."""
        if self.group not in ("SE3", "SE2", "SO3"):
            raise NotImplementedError(
                "Differentiating through the iterations (backward_mode='unroll' / 'truncated') is fused for SE3 / SE2 / SO3 pose "
                f"graphs (got {self.group}).  Use backward_mode='implicit', or call under torch.no_grad().")
        self.flush_variables()
        self.sync(force=True)

    def unrolled_step(self, opt, X: torch.Tensor, frozen: Optional[torch.Tensor], kwargs):
        """X -> (X exp(step * delta) where not ``frozen``, delta) as ONE autograd node over the kernels."""
        from .autograd import PGUnrolledIteration
        t = self.tensors
        return PGUnrolledIteration.apply(opt, self, frozen, kwargs, X, t.meas, t.w_between, t.prior_target, t.w_prior,
                                         t.log_radius_between, t.log_radius_prior)

    def where_state(self, mask: torch.Tensor, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """Per problem: ``a`` where ``mask`` else ``b`` (differentiable torch select on (P, B, ...) states)."""
        return torch.where(mask.view(1, -1, *([1] * (a.dim() - 2))), a, b)

    def state_with_graph(self, tensors):
        """The packed state assembled from the pose variables' OWN tensors (``optim_variables`` order), autograd history kept: where
        BackwardMode.UNROLL starts, so that gradients reach the values the caller passed in."""
        B = self.batch
        return torch.stack([t if t.shape[0] == B else t.expand(B, *t.shape[1:]) for t in tensors], dim=0)

    def unroll_incidence(self, device) -> torch.Tensor:
        """(P, Dmax) rows of [grad_pose_i (E) ; grad_pose_j (E) ; grad_pose_prior (K) ; zero row] that belong to each pose."""
        key = ("unroll_inc", str(device))
        if key not in self._scratch:
            s = self.structure
            E, Kp = s.num_edges, s.num_priors
            rows = [[] for _ in range(s.num_poses)]
            for e in range(E):
                rows[int(s.edge_i[e])].append(e)
                rows[int(s.edge_j[e])].append(E + e)
            for k in range(Kp):
                rows[int(s.prior_pose[k])].append(2 * E + k)
            dmax = max(1, max(len(r) for r in rows))
            pad = 2 * E + Kp
            self._scratch[key] = torch.tensor([r + [pad] * (dmax - len(r)) for r in rows], dtype=torch.long, device=device)
        return self._scratch[key]

    # ---- optimisation state: what the LM loop moves around (here: the packed pose buffer) --------------
    @property
    def state(self):
        return self.tensors.poses

    @property
    def device(self):
        return self.tensors.poses.device

    @property
    def optim_variables(self):
        return self.pose_vars

    def alloc_state(self):
        return torch.empty_like(self.tensors.poses)

    def clone_state(self):
        return self.tensors.poses.clone()

    def swap_state(self, new, repoint: bool = False):
        """Adopt ``new`` as the current state; returns the previous buffer for re-use."""
        old = self.tensors.poses
        self.set_poses(new, repoint=repoint)
        return old

    def keep_where(self, mask: torch.Tensor, out):
        """out <- current state where mask (B,), else out."""
        self.K.copy_where(mask, self.tensors.poses, out)

    def copy_where(self, mask: torch.Tensor, src, dst):
        """dst <- src where mask (B,): thx_copy_where, in place (torch.where(..., out=dst) with dst among the inputs
        allocates a state-sized temporary -- a hipMalloc stall of tens of ms at bundle-adjustment sizes)."""
        self.K.copy_where(mask, src, dst)

    def solution_dict(self, state):
        return {v.name: state[k].cpu() for k, v in enumerate(self.pose_vars)}

    def history_dict(self, hist, dtype):
        """``hist``: (K + 1, P, B, ...) packed states per iteration -> the reference's ``info.state_history`` layout
        (nonlinear_optimizer.py:150-163): name -> (B, ..., K + 1) on the host."""
        h = hist.to(dtype).cpu()
        return {v.name: h[:, k].movedim(0, -1).contiguous() for k, v in enumerate(self.pose_vars)}

    # ---- scratch ------------------------------------------------------------------------------
    def _buf(self, key, shape, dtype=None):
        t = self._scratch.get(key)
        dt = dtype or self.objective.dtype
        dev = self.tensors.poses.device
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dt or t.device != dev:
            t = torch.empty(*shape, dtype=dt, device=dev)
            self._scratch[key] = t
        return t

    @property
    def batch(self):
        return self.tensors.batch

    @property
    def dstruct(self):
        return self.structure.on(self.tensors.poses.device)

    # ---- fused operations ---------------------------------------------------------------------
    def assemble(self, H: torch.Tensor, g: torch.Tensor):
        self._refuse_fast_approx()
        self.sync()
        self.K.pg_assemble(self.dstruct, self.tensors, H, g)

    def supports_block_hessian(self) -> bool:
        """Block-compact storage of H (include/theseus_hip.h: thx_hblock_layout): SE3 pose graphs on the HIP kernels."""
        return self.group == "SE3" and hasattr(self.K, "pg_assemble_blocks")

    def assemble_blocks(self, Hc: torch.Tensor, g: torch.Tensor):
        self._refuse_fast_approx()
        self.sync()
        self.K.pg_assemble_blocks(self.dstruct, self.tensors, self.structure.hessian_blocks().on(Hc.device), Hc, g)

    def error_metric(self, poses: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, state=None):
        self.sync()
        poses = state if state is not None else poses
        B = self.batch
        part = self._buf("err_part", (ERR_CHUNKS, B))
        err = out if out is not None else torch.empty(B, dtype=self.objective.dtype, device=part.device)
        self.K.pg_error(self.dstruct, self.tensors, part, err, poses=poses)
        return err

    def retract(self, delta: torch.Tensor, step: float, ignore_mask: Optional[torch.Tensor], out: torch.Tensor):
        self.sync()
        m = None
        if ignore_mask is not None:
            m = ignore_mask if ignore_mask.dtype == torch.uint8 else ignore_mask.to(torch.uint8)
        self.K.retract(self.tensors.poses, delta, step, m, out)
        return out

    def jacobian_blocks(self, robust: bool = True, jacobians: bool = True):
        """Weighted Jacobian blocks / residuals of every cost: (J0,J1 (E,B,d,d), eb (E,B,d), Jp, ep), d = dof;
        robust costs rescaled as in ``weighted_jacobians_error`` unless ``robust=False``; ``jacobians=False`` writes the
        residuals only (J0 = J1 = Jp = None)."""
        self.sync()
        B, E, Kp, d = self.batch, self.structure.num_edges, self.structure.num_priors, self.dof
        dt, dev = self.objective.dtype, self.tensors.poses.device
        J0 = J1 = Jp = None
        if jacobians:
            J0 = torch.empty(max(E, 1), B, d, d, dtype=dt, device=dev)
            J1 = torch.empty_like(J0)
            Jp = torch.empty(max(Kp, 1), B, d, d, dtype=dt, device=dev)
        eb = torch.empty(max(E, 1), B, d, dtype=dt, device=dev)
        ep = torch.empty(max(Kp, 1), B, d, dtype=dt, device=dev)
        t = self.tensors if robust else self.tensors.without_robust()
        self.K.pg_jacobians(self.dstruct, t, J0, J1, eb, Jp, ep)
        if not jacobians:
            return None, None, eb[:E], None, ep[:Kp]
        return J0[:E], J1[:E], eb[:E], Jp[:Kp], ep[:Kp]

    def error_vector(self):
        """(B, m) weighted error in cost add order (Objective.error, core/objective.py:562-613)."""
        _, _, eb, _, ep = self.jacobian_blocks(robust=False, jacobians=False)
        t = self.tensors

        def robust_error(e, role_code, lr, table):
            # robust_cost_function.py:87-106: ones * sqrt(rho(|e|^2) / dim + eps), or sqrt(rho(e_r^2) + eps) per row with
            # flatten_dims -- elementwise torch on the device the errors live on (Objective.error() is not on the
            # optimiser's path)
            if not role_code:
                return e
            codes = table if table is not None else torch.full((e.shape[0],), role_code, dtype=torch.int32, device=e.device)
            codes = codes.view(-1, 1, 1)
            flat, kind = (codes & _lib.LOSS_FLATTEN) != 0, codes & ~_lib.LOSS_FLATTEN
            x = torch.where(flat, e ** 2, (e ** 2).sum(-1, keepdim=True).expand_as(e))
            r = lr.exp()
            welsch = r - r * torch.exp(-x / (r + 1e-20))
            huber = torch.where(x > r, 2 * torch.sqrt(r * torch.maximum(x, r) + 1e-20) - r, x)
            hinge = torch.where(x > r, x.sqrt() - r.sqrt(), torch.full_like(x, 1e-20))
            gm = r * x / (r + x + 1e-20)
            rho = torch.where(kind == _lib.LOSS_WELSCH, welsch, torch.where(kind == _lib.LOSS_HINGE, hinge,
                                                                         torch.where(kind == _lib.LOSS_GEMAN_MCCLURE, gm, huber)))
            h = torch.where(flat, (rho + 1e-20).sqrt(), (rho / e.shape[-1] + 1e-20).sqrt())
            return torch.where(kind == _lib.LOSS_NONE, e, h)
        eb = robust_error(eb, t.robust_between, t.log_radius_between, t.loss_between)
        ep = robust_error(ep, t.robust_prior, t.log_radius_prior, t.loss_prior)
        B = self.batch
        out = torch.empty(B, self.m, dtype=eb.dtype, device=eb.device)
        s = self.structure
        if s.num_edges:
            rows = torch.from_numpy(s.edge_row_start).to(eb.device)
            idx = (rows.view(-1, 1) + torch.arange(self.dof, device=eb.device)).view(-1)
            out[:, idx] = eb.permute(1, 0, 2).reshape(B, -1)
        if s.num_priors:
            rows = torch.from_numpy(s.prior_row_start).to(eb.device)
            idx = (rows.view(-1, 1) + torch.arange(self.dof, device=eb.device)).view(-1)
            out[:, idx] = ep.permute(1, 0, 2).reshape(B, -1)
        return out


    def jacobian_times(self, blocks, v: torch.Tensor) -> torch.Tensor:
        """A v (B, m) from the weighted Jacobian BLOCKS of ``jacobian_blocks()`` -- the product the reference takes with its
        dense (B, m, n) Jacobian (dense_linearization.py:73-74), which at the headline size would be 155 GB.  Rows in cost
        add order, columns in the packed (= the linearization's) variable order."""
        J0, J1, _, Jp, _ = blocks
        s, d, B = self.structure, self.dof, v.shape[0]
        vb = v.reshape(B, -1, d)
        out = torch.zeros(B, self.m, dtype=v.dtype, device=v.device)
        dev = v.device
        if s.num_edges:
            vi = vb[:, torch.from_numpy(s.edge_i).long().to(dev)].permute(1, 0, 2).unsqueeze(-1)   # (E, B, d, 1)
            vj = vb[:, torch.from_numpy(s.edge_j).long().to(dev)].permute(1, 0, 2).unsqueeze(-1)
            r = (J0 @ vi + J1 @ vj).squeeze(-1)                                                    # (E, B, d)
            rows = torch.from_numpy(s.edge_row_start).to(dev)
            idx = (rows.view(-1, 1) + torch.arange(d, device=dev)).view(-1)
            out[:, idx] = r.permute(1, 0, 2).reshape(B, -1)
        if s.num_priors:
            vp = vb[:, torch.from_numpy(s.prior_pose).long().to(dev)].permute(1, 0, 2).unsqueeze(-1)
            r = (Jp @ vp).squeeze(-1)
            rows = torch.from_numpy(s.prior_row_start).to(dev)
            idx = (rows.view(-1, 1) + torch.arange(d, device=dev)).view(-1)
            out[:, idx] = r.permute(1, 0, 2).reshape(B, -1)
        return out


def packed_for(objective: Objective, kernels=None, order=None):
    """Get (or build) the packed representation attached to an objective (``order``: variable names in column order): the fused
    pose-graph one, or -- Euclidean variables with cost functions that hand over their Jacobian blocks -- the generic one of
    theseus_amd/euclidean.py."""
    from .euclidean import PackedEuclidean
    p = getattr(objective, "_packed", None)
    order = tuple(order) if order is not None else tuple(objective.optim_vars.keys())
    if (p is None or not isinstance(p, (PackedPoseGraph, PackedEuclidean)) or p.version != objective.current_version
            or (kernels is not None and p.K is not kernels) or p.order != order):
        try:
            p = PackedPoseGraph(objective, kernels, order)
        except UnsupportedObjective as fused_error:
            if not isinstance(objective, Objective):   # the reference's own Objective: theseus_amd/plugin.py has its generic path
                raise
            try:
                p = PackedEuclidean(objective, kernels, order)
            except UnsupportedObjective:
                raise fused_error from None
        objective._packed = p
    return p
