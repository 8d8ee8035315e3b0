"""Gauss-Newton / Levenberg-Marquardt outer loop (host side) driving the HIP back end.

Mirrors, call for call, what the reference loop asks of the Linearization / LinearSolver plugin
surface:
  NonlinearOptimizer / Info / Status  theseus/optimizer/nonlinear/nonlinear_optimizer.py:39-294
  NonlinearLeastSquares loop, _step   theseus/optimizer/nonlinear/nonlinear_least_squares.py:58-380
  GaussNewton                         theseus/optimizer/nonlinear/gauss_newton.py
  LevenbergMarquardt                  theseus/optimizer/nonlinear/levenberg_marquardt.py:50-201
The loop state (poses, errors, damping, masks, histories) lives in device memory and the step is
five kernel launches (assemble, factor, solve, retract, error) + one for LM's accept test; the
host synchronises at most once per iteration, and not at all when both tolerances are 0 and damping
is not adaptive (the three batch-global predicates of the reference cannot fire then).
"""
import abc
import warnings
from dataclasses import dataclass
from enum import Enum
from typing import Any, Callable, Dict, Optional, Type, Union

import numpy as np
import torch

from .core import Objective
from .linear_solver import HipCholeskySolver, LinearSolver
from .linearization import HipLinearization, Linearization
from .sharding import LocalBatchReducer


class NonlinearOptimizerStatus(Enum):
    START = 0
    CONVERGED = 1
    MAX_ITERATIONS = 2
    FAIL = -1


class BackwardMode(Enum):
    UNROLL = 0
    IMPLICIT = 1
    TRUNCATED = 2
    DLM = 3

    @staticmethod
    def resolve(key: Union[str, "BackwardMode"]) -> "BackwardMode":
        if isinstance(key, BackwardMode):
            return key
        if not isinstance(key, str):
            raise ValueError("Backward mode must be th.BackwardMode or string.")
        try:
            return BackwardMode[key.upper()]
        except KeyError:
            raise ValueError(f"Unrecognized backward mode f{key}. Valid choices are unroll, implicit, truncated, dlm.")


@dataclass
class NonlinearOptimizerInfo:
    best_solution: Optional[Dict[str, torch.Tensor]]
    status: np.ndarray
    converged_iter: torch.Tensor
    best_iter: torch.Tensor
    err_history: Optional[torch.Tensor]
    last_err: torch.Tensor
    best_err: torch.Tensor
    iters_done: int = 0
    state_history: Optional[Dict[str, torch.Tensor]] = None  # variable name -> (B, ..., max_iterations + 1)


@dataclass
class NonlinearOptimizerParams:
    abs_err_tolerance: float
    rel_err_tolerance: float
    max_iterations: int
    step_size: float


class _LaggedFlag:
    """A device-side bool looked at WITHOUT stalling the queue: ``post`` copies it to pinned host memory behind the work
    already queued and records an event; ``seen`` is True once a posted copy has landed with the flag raised.  On a CPU
    device (tests) the look is immediate."""

    def __init__(self, device):
        self.cuda = device.type == "cuda"
        self.pending, self.value = [], False
        self.host = torch.zeros(8, dtype=torch.bool).pin_memory() if self.cuda else None
        self.slot = 0

    def post(self, flag: torch.Tensor):
        if not self.cuda:
            self.value = self.value or bool(flag)
            return
        k = self.slot % 8
        self.slot += 1
        self.host[k:k + 1].copy_(flag.view(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((ev, k))

    def seen(self) -> bool:
        while self.pending and self.pending[0][0].query():
            _, k = self.pending.pop(0)
            self.value = self.value or bool(self.host[k])
        return self.value


class _StateHistory:
    """``track_state_history`` (nonlinear_optimizer.py:150-163,174-178): the iterate after every counted iteration.  The states
    are kept on the DEVICE in the packed layout -- slot k of a (K + 1, ...) buffer per state tensor, written at a device-side
    index so that the sync-free loop stays sync-free -- and turned into the reference's name -> (B, ..., K + 1) host tensors
    once, after the loop.  Slots never reached stay at inf, like the reference's."""

    def __init__(self, state, slots: int):
        self.tuple = isinstance(state, tuple)
        ts = state if self.tuple else (state,)
        self.bufs = tuple(torch.full((slots,) + tuple(t.shape), float("inf"), dtype=t.dtype, device=t.device) for t in ts)
        self.slots = slots
        self.record(0, state)

    def record(self, slot, state, counted=None):
        """``slot``: int, or a 0-dim device tensor (clamped to the buffer); ``counted``: 0-dim device bool -- write only if set."""
        ts = state if self.tuple else (state,)
        for buf, t in zip(self.bufs, ts):
            t = t.detach()
            if isinstance(slot, int):
                buf[slot].copy_(t)
                continue
            idx = slot.clamp(max=self.slots - 1).view(1)
            if counted is not None:
                t = torch.where(counted, t, buf.index_select(0, idx)[0])
            buf.index_copy_(0, idx, t.unsqueeze(0))

    def cut(self, first_unused: int):
        for buf in self.bufs:
            buf[first_unused:] = float("inf")

    def to_dict(self, packed):
        hist = self.bufs if self.tuple else self.bufs[0]
        return packed.history_dict(hist, torch.get_default_dtype())   # (the reference allocates it with torch.ones(...): default dtype)


class NonlinearLeastSquares(abc.ABC):
    _MAX_ALL_REJECT_ATTEMPTS = 3  # nonlinear_optimizer.py:88

    def __init__(self, objective: Objective, linear_solver_cls: Optional[Type[LinearSolver]] = None,
                 vectorize: bool = False, linearization_cls: Optional[Type[Linearization]] = None,
                 linearization_kwargs: Optional[Dict[str, Any]] = None,
                 linear_solver_kwargs: Optional[Dict[str, Any]] = None, abs_err_tolerance: float = 1e-10,
                 rel_err_tolerance: float = 1e-8, max_iterations: int = 20, step_size: float = 1.0, **kwargs):
        self.objective = objective
        if linear_solver_cls is None:
            # bundle-adjustment objectives (SE3 cameras + Point3 points) default to the Schur-complement solver
            from .ba import HipSchurSolver
            kinds = {type(v).__name__ for v in objective.optim_vars.values()}
            linear_solver_cls = HipSchurSolver if {"Point3", "SE3"} <= kinds else HipCholeskySolver
        self.linear_solver = linear_solver_cls(objective, linearization_cls=linearization_cls,
                                               linearization_kwargs=linearization_kwargs,
                                               **(linear_solver_kwargs or {}))
        self.ordering = self.linear_solver.linearization.ordering
        self.params = NonlinearOptimizerParams(abs_err_tolerance, rel_err_tolerance, max_iterations, step_size)
        self.reducer = LocalBatchReducer()
        self._objective_version = objective.current_version

    def set_params(self, **kwargs):
        for k, v in kwargs.items():
            if not hasattr(self.params, k):
                raise ValueError(f"Invalid nonlinear optimizer parameter {k}.")
            setattr(self.params, k, v)

    # ---- hooks for subclasses ------------------------------------------------------------------
    def reset(self, **kwargs):
        self.linear_solver.reset(**kwargs)

    @abc.abstractmethod
    def compute_delta(self, **kwargs) -> torch.Tensor:
        pass

    def _complete_step(self, delta, new_err, previous_err, **kwargs) -> Optional[torch.Tensor]:
        return None

    def _may_reject(self, kwargs) -> bool:
        """Can ``_complete_step`` reject steps with these optimizer kwargs?  A subclass that overrides ``_complete_step`` may
        reject unless it says otherwise."""
        return type(self)._complete_step is not NonlinearLeastSquares._complete_step

    def _join_compute_delta(self):
        """Called INSTEAD of ``compute_delta`` on a rank that cannot run it (host-side error): take part in whatever
        collectives ``compute_delta`` issues, so that the other shards are not left waiting."""

    # ---- public entry (theseus/optimizer/optimizer.py:40-53) -------------------------------------
    def optimize(self, **kwargs) -> NonlinearOptimizerInfo:
        if self._objective_version != self.objective.current_version:
            raise RuntimeError("The objective was modified after optimizer construction, which is "
                               "currently not supported.")
        return self._optimize_impl(**kwargs)

    # nonlinear_optimizer.py:274-294
    def _split_backward_iters(self, backward_mode=BackwardMode.UNROLL, backward_num_iterations=None, **kw):
        if backward_mode in (BackwardMode.UNROLL, BackwardMode.DLM):
            return self.params.max_iterations, 0
        if backward_mode == BackwardMode.IMPLICIT:
            return 1, self.params.max_iterations - 1
        if backward_num_iterations is None:
            raise ValueError("backward_num_iterations expected but not received.")
        return backward_num_iterations, self.params.max_iterations - backward_num_iterations

    def _check_convergence(self, err, last_err):
        """nonlinear_optimizer.py:110-119 (the mean-test is reduced across shards by self.reducer)."""
        change = last_err - err
        conv = (change.abs() < self.params.abs_err_tolerance) | (
            (change / last_err).abs() < self.params.rel_err_tolerance)
        return conv

    def _optimize_impl(self, **kwargs) -> NonlinearOptimizerInfo:
        """The loop moves the state through private, recycled buffers and re-points the Variable objects only at the end
        (``_vars_stale``).  If anything raises in between -- bad optimizer kwargs, a kernel error, KeyboardInterrupt -- the
        variables are re-pointed at the state reached so far BEFORE the exception leaves: otherwise the next
        ``forward(new_inputs)`` would see the stale flag, flush the old private buffer over the user's new tensors and
        silently optimise the previous problem."""
        packed = self.linear_solver.linearization.packed
        try:
            return self._optimize_loop(**kwargs)
        except BaseException:
            with torch.no_grad():
                packed.flush_variables()   # (a no-op unless the loop left the variables pointing at a stale buffer)
            raise
        finally:
            packed._keep_graph_tensors = False

    def _optimize_loop(self, track_best_solution: bool = False, track_err_history: bool = False,
                       track_state_history: bool = False, verbose: bool = False,
                       backward_mode: Union[str, BackwardMode] = BackwardMode.UNROLL,
                       end_iter_callback: Optional[Callable] = None, **kwargs) -> NonlinearOptimizerInfo:
        backward_mode = BackwardMode.resolve(backward_mode)
        outer_grad = torch.is_grad_enabled()
        if backward_mode == BackwardMode.DLM:
            raise NotImplementedError("backward_mode=DLM is not supported by the HIP back end "
                                      "(supported: 'implicit'; 'unroll' / 'truncated' see below).")
        implicit = backward_mode == BackwardMode.IMPLICIT
        lin: HipLinearization = self.linear_solver.linearization
        packed = lin.packed
        # Differentiating THROUGH iterations (UNROLL: all of them, TRUNCATED: the last ``backward_num_iterations``;
        # nonlinear_least_squares.py:222-282): the Hessian is part of the graph there, i.e. the derivatives of every cost's
        # Jacobian are needed.  The generic path has them (its blocks come from torch: theseus_amd/euclidean.py); the fused
        # pose-graph / bundle-adjustment paths have one autograd node per iteration (thx_pg*_unroll_vjp / thx_ba_unroll_vjp).
        # (UNROLL differentiates from the FIRST iteration: the caller's own tensors of the optimisation variables -- a learned
        #  initialisation passed through TheseusLayer.forward -- are part of the graph; kept before the first sync re-points the variables)
        init_tensors = None
        if backward_mode in (BackwardMode.UNROLL, BackwardMode.TRUNCATED) and outer_grad:   # (TRUNCATED: only if its no_grad head is empty)
            ts = [v.tensor for v in packed.optim_variables]
            if any(t.requires_grad for t in ts):
                init_tensors = ts
        unrolled = (backward_mode in (BackwardMode.UNROLL, BackwardMode.TRUNCATED) and outer_grad
                    and (self._needs_grad() or init_tensors is not None))
        if unrolled and not hasattr(packed, "unrolled_step"):
            raise NotImplementedError(
                f"Differentiating through the iterations (backward_mode='{backward_mode.name.lower()}') needs a packer with "
                f"unrolled_step() (generic, pose-graph and bundle-adjustment objectives have one; got {type(packed).__name__}).  "
                "Use backward_mode='implicit', or call under torch.no_grad().")
        if unrolled and isinstance(self, TrustRegion):
            raise NotImplementedError("differentiable iterations (backward_mode='unroll' / 'truncated' with gradients): Gauss-Newton / "
                                      "Levenberg-Marquardt (a trust-region step reads Av / the Cauchy point outside autograd).")
        # (only THEN may a no_grad re-pack leave the variables on their own graph-carrying tensors instead of views of the packed state)
        packed._keep_graph_tensors = init_tensors is not None
        with torch.no_grad():
            packed._defer_repoint = True   # (PackedPoseGraph.sync: the variables are re-pointed once, when the loop is done)
            try:
                packed.sync(deep=True)   # once per optimize(): also catches in-place edits of the variables' tensors
            finally:
                packed._defer_repoint = False
        self.reset(**kwargs, backward_mode=backward_mode)
        tail_iters = 0
        if implicit:
            _, loop_iters = self._split_backward_iters(backward_mode=backward_mode, **kwargs)
        elif backward_mode == BackwardMode.TRUNCATED:
            # (without anything to differentiate the two loops of the reference are one loop with a fresh convergence mask in
            #  between; run as the differentiable tail only when there is a graph to build)
            tail, head = self._split_backward_iters(backward_mode=backward_mode, **kwargs)
            loop_iters, tail_iters = (head, tail) if unrolled else (self.params.max_iterations, 0)
        else:
            loop_iters, tail_iters = (0, self.params.max_iterations) if unrolled else (self.params.max_iterations, 0)
        with torch.no_grad():
            packed.sync()
            packed.privatize_state()   # never recycle the buffer the previous optimize() handed to the user
            B = packed.batch
            dev, dt = packed.device, self.objective.dtype
            p = self.params
            need_conv = p.abs_err_tolerance > 0 or p.rel_err_tolerance > 0
            adaptive = bool(kwargs.get("adaptive_damping", False))
            inf = float("inf")

            def fresh_info():
                last = packed.error_metric()
                hist = None
                if track_err_history:
                    hist = torch.full((B, p.max_iterations + 1), inf, dtype=dt, device=dev)
                    hist[:, 0] = last
                return last, hist, NonlinearOptimizerInfo(
                    best_solution=None, status=np.array([NonlinearOptimizerStatus.START] * B),
                    converged_iter=torch.full((B,), -1, dtype=torch.long), best_iter=torch.zeros(B, dtype=torch.long),
                    err_history=hist, last_err=last, best_err=last.clone())

            last_err, err_hist, info = fresh_info()
            state_hist = _StateHistory(packed.state, p.max_iterations + 1) if track_state_history else None
            if track_best_solution:
                best_state = packed.clone_state()
                best_err = last_err.clone()
                best_iter = torch.zeros(B, dtype=torch.long, device=dev)
            if verbose:
                print(f"Nonlinear optimizer. Iteration: 0. Error: {last_err.mean().item()}")

            converged = None          # (B,) bool on device, None == nobody
            conv_iter = torch.full((B,), -1, dtype=torch.long, device=dev)
            spare = packed.alloc_state()
            err_new = torch.empty(B, dtype=dt, device=dev)
            it, all_reject_attempts = 0, 0

            def warn_failed(run_err):
                warnings.warn(f"There was an error while running the linear optimizer. Original error message: {run_err}.",
                              RuntimeWarning)

            # ---- SYNC-FREE iterations.  The host decisions of the reference's iteration -- "did a linear
            #      solve fail?" (nonlinear_least_squares.py:138-152), "were ALL steps rejected?" (:358-359, the retry that
            #      does not count as an iteration), "has EVERY problem converged?" (:202) -- are evaluated on the device
            #      (all-reduced over the shards when the batch is sharded) and kept there as three flags; everything else
            #      of the iteration is per-problem masked work: rejected problems keep their state and error
            #      (thx_copy_where), converged / failed ones are frozen through the retraction mask, lambda lives in a device
            #      vector (thx_lm_accept).  The iterations are queued back to back and the flags are read ONCE after the loop:
            #        * a failed solve froze every later update on the device: FAIL status, iters_done = the failing iteration;
            #        * everybody converged at iteration k: later iterations were no-ops on frozen problems, the bookkeeping
            #          is cut at k (a lagged, non-blocking poll of the flag stops the queueing a couple of iterations later);
            #        * an ALL-rejected step is the reference's uncounted retry: the iteration counter itself lives on the
            #          device (see below), nothing is replayed.
            #      Callbacks and verbose printing need the host every iteration: synchronous path. ----
            sync_free = (isinstance(self.reducer, LocalBatchReducer) and end_iter_callback is None and not verbose
                         and loop_iters > 0)
            halt = False
            if sync_free:
                # What the host queues are ATTEMPTS.  An all-rejected attempt leaves every problem's state and error where they
                # were and has already moved lambda / the trust region -- which is precisely the reference's retry
                # (nonlinear_least_squares.py:358-359: ``continue`` without counting the iteration; the third all-rejected attempt
                # in a row is counted with err = last_err, nonlinear_optimizer.py:88).  So nothing is replayed: the ITERATION
                # COUNTER lives on the device (``it_dev``), the histories are written at device-side column indices, and after the
                # queued attempts the host reads the counter and queues what is missing (no all-rejection: one round, one sync).
                flag = lambda v: torch.full((), v, dtype=torch.long, device=dev)  # noqa: E731
                failed = torch.zeros((), dtype=torch.bool, device=dev)
                first_fail, first_all_conv = flag(-1), flag(-1)
                it_dev, attempts_dev = flag(0), flag(0)
                true0 = torch.ones((), dtype=torch.bool, device=dev)
                # (the lagged poll is a per-process decision: a sharded batch runs all its attempts, so that every rank
                #  issues the same all-reduces)
                poll = _LaggedFlag(dev) if (need_conv and self.reducer.world_size == 1) else None
                raised = None
                while True:
                    for _ in range(loop_iters - it):
                        if poll is not None and poll.seen():
                            break   # everybody converged a moment ago: stop queueing (the cut below is exact)
                        local_fail = None
                        if raised is None:
                            lin.linearize()
                            try:
                                delta = self.compute_delta(**kwargs)
                                local_fail = self.linear_solver.info.ne(0).any()
                            except RuntimeError as run_err:
                                # a host-side error on THIS rank: the other shards are (or will be) waiting in device_any()'s
                                # all-reduce -- keep taking part in it with the flag raised instead of leaving the loop
                                raised = run_err
                        if raised is not None:
                            self._join_compute_delta()
                            delta = torch.zeros(B, lin.num_cols, dtype=dt, device=dev)
                            local_fail = torch.ones((), dtype=torch.bool, device=dev)
                        now = self.reducer.device_any(local_fail)
                        first_fail = torch.where(now & ~failed, it_dev, first_fail)
                        failed = failed | now
                        stop = failed | it_dev.ge(loop_iters)        # nothing moves after a failure / beyond the last iteration
                        frozen = stop.expand(B) if converged is None else (converged | stop)
                        packed.retract(delta, p.step_size, frozen.to(torch.uint8).contiguous(), spare)
                        packed.error_metric(state=spare, out=err_new)
                        reject = self._complete_step(delta, err_new, last_err, step_size=p.step_size, **kwargs)
                        counted = ~stop
                        if reject is not None:
                            rb = reject.bool()
                            all_rej = self.reducer.device_all(rb.all())
                            forced = attempts_dev.ge(self._MAX_ALL_REJECT_ATTEMPTS - 1)
                            counted = counted & (~all_rej | forced)
                            attempts_dev = torch.where(all_rej & ~forced & ~stop, attempts_dev + 1, torch.zeros_like(attempts_dev))
                            packed.keep_where(rb, spare)                                   # rejected problems keep their state
                            packed.K.copy_where(rb, last_err.view(1, -1, 1), err_new.view(1, -1, 1))   # ... and their error
                        err = torch.where(stop, last_err, err_new)
                        spare = packed.swap_state(spare)
                        if err_hist is not None:
                            col = (it_dev + 1).clamp(max=p.max_iterations).view(1)
                            cur = err_hist.index_select(1, col)
                            err_hist.index_copy_(1, col, torch.where(counted, err.unsqueeze(1), cur))
                        if state_hist is not None:
                            state_hist.record(it_dev + 1, packed.state, counted)
                        if track_best_solution:
                            better = (err < best_err) & counted
                            packed.copy_where(better, packed.state, best_state)
                            best_err = torch.where(better, err, best_err)
                            best_iter = torch.where(better, it_dev, best_iter)  # nonlinear_optimizer.py:202
                        if need_conv:
                            small = self.reducer.device_mean_abs_below(err, p.abs_err_tolerance)
                            conv_now = self._check_convergence(err, last_err) | small
                            # (an uncounted attempt -- all rejected, err == last_err everywhere -- converges nobody)
                            converged = (conv_now & counted) if converged is None else torch.where(counted, conv_now, converged)
                            conv_iter = torch.where(converged & (conv_iter < 0) & counted, it_dev + 1, conv_iter)
                            all_conv = self.reducer.device_all(converged.all()) & counted
                            first_all_conv = torch.where(all_conv & (first_all_conv < 0), it_dev + 1, first_all_conv)
                            if poll is not None:
                                poll.post(first_all_conv >= 0)
                        last_err = err
                        it_dev = it_dev + counted.long()
                    # ---- the one host sync of the round ----
                    f_fail, f_conv, it_now = torch.stack([first_fail, first_all_conv, it_dev]).tolist()
                    if f_fail >= 0 and not (f_conv >= 0 and f_conv < f_fail + 1):
                        try:
                            if raised is not None:
                                raise raised
                            self.linear_solver.check_info()
                            raise RuntimeError("the linear solve failed on another shard of the batch")
                        except RuntimeError as run_err:
                            warn_failed(run_err)
                        info.status[:] = NonlinearOptimizerStatus.FAIL   # (overrides every status: nonlinear_least_squares.py:147)
                        it = f_fail
                        halt = True
                    elif f_conv >= 0:
                        # the reference broke out of its loop here, BEFORE counting the converging iteration (:202-203): its error
                        # is in err_history[:, f_conv], the iteration count stays at f_conv - 1 (the synchronous path below and
                        # the implicit epilogue -- which writes its error at [iters_done + 1], as _merge_infos does -- agree)
                        it = f_conv - 1
                        if err_hist is not None:
                            err_hist[:, f_conv + 1:] = inf
                        if state_hist is not None:
                            state_hist.cut(f_conv + 1)
                        converged = conv_iter.ge(0) & conv_iter.le(f_conv)   # (later iterations only re-marked frozen problems)
                        conv_iter = torch.where(converged, conv_iter, torch.full_like(conv_iter, -1))
                        halt = True
                    else:
                        it = it_now
                    if halt or it >= loop_iters:
                        break
                info.last_err = last_err   # (frozen / failed problems kept their error: this is the error AT ``it``)
                info.iters_done = it
            else:
                # ---- synchronous path (callbacks, verbose): one host decision per iteration ----
                while it < loop_iters:
                    lin.linearize()
                    try:
                        delta = self.compute_delta(**kwargs)
                    except RuntimeError as run_err:
                        if self.reducer.world_size > 1:
                            # the other shards are (or will be) waiting in compute_delta's collectives and in this iteration's
                            # decide(): take part in both with the failure flag raised, so that EVERY rank leaves the loop here
                            self._join_compute_delta()
                            one = torch.ones(1, dtype=torch.bool, device=dev)
                            rej = [one] if self._may_reject(kwargs) else []
                            self.reducer.decide([one] + rej, rej)
                        warn_failed(run_err)
                        info.status[:] = NonlinearOptimizerStatus.FAIL
                        halt = True
                        break
                    # retract (converged problems frozen) + error of the candidate, fused HIP kernels
                    packed.retract(delta, p.step_size, converged, spare)
                    packed.error_metric(state=spare, out=err_new)
                    reject = self._complete_step(delta, err_new, last_err, step_size=p.step_size, **kwargs)
                    # ---- the only host sync of the iteration: [solver failed | all rejected, any rejected], over
                    #      the GLOBAL batch (self.reducer all-reduces across shards when the batch is sharded) ----
                    any_f = [self.linear_solver.info.ne(0)]
                    all_f = []
                    if reject is not None:
                        rb = reject.bool()
                        any_f.append(rb)
                        all_f.append(rb)
                    any_r, all_r = self.reducer.decide(any_f, all_f)
                    if any_r[0]:
                        try:
                            self.linear_solver.check_info()
                            raise RuntimeError("the linear solve failed on another shard of the batch")
                        except RuntimeError as run_err:
                            warn_failed(run_err)
                        info.status[:] = NonlinearOptimizerStatus.FAIL
                        halt = True
                        break
                    if reject is not None:
                        all_rej, any_rej = all_r[0], any_r[1]
                        if all_rej:
                            all_reject_attempts += 1
                            if all_reject_attempts < self._MAX_ALL_REJECT_ATTEMPTS:
                                continue
                            err = last_err
                        else:
                            if any_rej:
                                rb = reject.bool()
                                packed.keep_where(rb, spare)
                                packed.K.copy_where(rb, last_err.view(1, -1, 1), err_new.view(1, -1, 1))
                                err = err_new.clone()
                            else:
                                err = err_new.clone()
                            spare = packed.swap_state(spare)
                    else:
                        err = err_new.clone()
                        spare = packed.swap_state(spare)
                    all_reject_attempts = 0
                    if err_hist is not None:
                        err_hist[:, it + 1] = err
                    if state_hist is not None:
                        state_hist.record(it + 1, packed.state)
                    if track_best_solution:
                        better = err < best_err
                        packed.copy_where(better, packed.state, best_state)
                        best_err = torch.where(better, err, best_err)
                        best_iter = torch.where(better, torch.full_like(best_iter, it), best_iter)  # nonlinear_optimizer.py:202
                    if verbose:
                        print(f"Nonlinear optimizer. Iteration: {it + 1}. Error: {err.mean().item()}")
                    if need_conv:
                        if self.reducer.mean_abs(err) < p.abs_err_tolerance:
                            converged = torch.ones(B, dtype=torch.bool, device=dev)
                        else:
                            converged = self._check_convergence(err, last_err)
                        conv_iter = torch.where(converged & (conv_iter < 0), torch.full_like(conv_iter, it + 1), conv_iter)
                        if self.reducer.decide([], [converged])[1][0]:
                            info.last_err = err
                            halt = True
                            break
                    last_err = err
                    info.last_err = err
                    if end_iter_callback is not None:
                        packed.flush_variables()
                        end_iter_callback(self, info, delta, it)
                    it += 1
                    info.iters_done = it

            # ---- BackwardMode.UNROLL / TRUNCATED with a graph: the reference's SECOND loop (nonlinear_least_squares.py:264-282)
            #      -- same iteration, caller's grad mode, a fresh convergence mask -- then _merge_infos
            #      (nonlinear_optimizer.py:220-266).  Host-synchronous: these are a handful of small iterations. ----
            if tail_iters > 0 and not (info.status == NonlinearOptimizerStatus.FAIL).any():
                packed.flush_variables()
                with torch.set_grad_enabled(outer_grad):
                    packed.prepare_unroll()
                det = lambda x: tuple(y.detach() for y in x) if isinstance(x, tuple) else x.detach()  # noqa: E731  (BA: (cams, points))
                X = det(packed.state)
                if init_tensors is not None and it == 0 and hasattr(packed, "state_with_graph"):
                    with torch.set_grad_enabled(outer_grad):
                        X = packed.state_with_graph(init_tensors)     # same values, the caller's autograd history
                g_conv = torch.zeros(B, dtype=torch.bool, device=dev)
                g_conv_iter = torch.zeros(B, dtype=torch.long, device=dev)   # the reference's counter: += 1 while not converged
                g_status_conv = torch.zeros(B, dtype=torch.bool, device=dev)
                g_last, g_it, attempts, g_errs = last_err, 0, 0, []
                while g_it < tail_iters:
                    with torch.set_grad_enabled(outer_grad):
                        X_cand, delta = packed.unrolled_step(self, X, g_conv, kwargs)
                    err_new = packed.error_metric(state=det(X_cand))
                    reject = self._complete_step(delta.detach(), err_new, g_last, step_size=p.step_size, **kwargs)
                    if reject is not None:
                        rb = reject.bool()
                        if bool(rb.all()):
                            attempts += 1
                            if attempts < self._MAX_ALL_REJECT_ATTEMPTS:
                                continue
                        with torch.set_grad_enabled(outer_grad):
                            X_cand = packed.where_state(rb, X, X_cand)
                        err = torch.where(rb, g_last, err_new)
                    else:
                        err = err_new
                    attempts = 0
                    X = X_cand
                    # _update_info, THEN _check_convergence (nonlinear_least_squares.py:192-203): the iteration on which everybody
                    # converges is not counted, but it has been through _update_info -- it competes for best_solution, and in UNROLL's
                    # single loop its error / state are in the histories at [it + 1].  TRUNCATED's second loop has its own info, of
                    # which _merge_infos (nonlinear_optimizer.py:220-266) copies ``grad_iters_done`` history columns (not the
                    # converging iteration's), merges best_solution / best_err, and leaves best_iter the first loop's.
                    single_loop = backward_mode == BackwardMode.UNROLL
                    g_conv_iter = g_conv_iter + (~g_conv).long()
                    if verbose:
                        print(f"Nonlinear optimizer. Iteration: {it + g_it + 1}. Error: {err.mean().item()}")
                    conv_new, everybody = g_conv, False
                    if need_conv:
                        conv_new = (torch.ones_like(g_conv) if self.reducer.mean_abs(err) < p.abs_err_tolerance
                                    else self._check_convergence(err, g_last))
                        everybody = bool(conv_new.all())
                    if single_loop or not everybody:
                        g_errs.append(err)
                        if state_hist is not None:         # (detached copies, nonlinear_optimizer.py:150-207)
                            state_hist.record(it + g_it + 1, det(X))
                    if track_best_solution:
                        better = err < best_err
                        packed.copy_where(better, det(X), best_state)
                        best_err = torch.where(better, err, best_err)
                        if single_loop:
                            best_iter = torch.where(better, torch.full_like(best_iter, it + g_it), best_iter)
                    if need_conv:
                        g_conv = conv_new
                        g_status_conv = g_status_conv | g_conv
                        if everybody:
                            break
                    g_last = err
                    if end_iter_callback is not None:
                        packed.swap_state(X, repoint=True)
                        end_iter_callback(self, info, delta.detach(), it + g_it)
                    g_it += 1
                packed.swap_state(X, repoint=True)      # the variables view the (graph-carrying) result
                # _merge_infos
                if err_hist is not None and g_errs:
                    err_hist[:, it + 1:it + 1 + len(g_errs)] = torch.stack(g_errs, 1)   # (UNROLL: the converging iteration's too)
                undecided = ~(converged if converged is not None else torch.zeros(B, dtype=torch.bool, device=dev))
                if need_conv:
                    # problems the first loop left at MAX_ITERATIONS take the second loop's verdict; their counters add up
                    first_count = torch.full_like(conv_iter, it)
                    newly = undecided & g_status_conv
                    conv_iter = torch.where(newly, first_count + g_conv_iter, conv_iter)
                    converged = newly if converged is None else (converged | newly)
                # TRUNCATED: _merge_infos leaves the FIRST loop's last_err in the merged info (nonlinear_optimizer.py:220-266);
                # UNROLL is ONE loop in the reference (nonlinear_least_squares.py:252-268), whose _update_info sets it every iteration
                info.last_err = g_last.detach() if backward_mode == BackwardMode.UNROLL else last_err
                it += g_it
                info.iters_done = it

            # ---- BackwardMode.IMPLICIT: the last step is an undamped Gauss-Newton step with the Hessian detached,
            #      executed under the caller's grad mode (nonlinear_least_squares.py:121-135,265-292) ----
            if implicit and not (info.status == NonlinearOptimizerStatus.FAIL).any():
                X_new, delta = self._implicit_last_step(packed, outer_grad, kwargs)
                X_det = tuple(x.detach() for x in X_new) if isinstance(X_new, tuple) else X_new.detach()
                err = packed.error_metric(state=X_det)
                packed.swap_state(X_new)
                if err_hist is not None:
                    err_hist[:, it + 1] = err
                if state_hist is not None:
                    state_hist.record(it + 1, X_det)
                if track_best_solution:
                    better = err < best_err
                    packed.copy_where(better, X_det, best_state)
                    best_err = torch.where(better, err, best_err)
                    best_iter = torch.where(better, torch.full_like(best_iter, it), best_iter)  # nonlinear_optimizer.py:202
                if need_conv:
                    # _merge_infos (nonlinear_least_squares.py:216-263): a problem that converged in the no-grad loop stays
                    # CONVERGED whatever the final Gauss-Newton step does
                    last_step = self._check_convergence(err, last_err)
                    converged = last_step if converged is None else (converged | last_step)
                    conv_iter = torch.where(converged & (conv_iter < 0), torch.full_like(conv_iter, it + 1), conv_iter)
                last_err = err
                info.last_err = err
                if end_iter_callback is not None:
                    packed.flush_variables()
                    end_iter_callback(self, info, delta.detach(), it)
                it += 1
                info.iters_done = it
            packed.flush_variables()
            # ---- bookkeeping, one device->host copy ----
            if getattr(self.linear_solver, "_check_singular", False) and hasattr(self.linear_solver, "post_singular_warning"):
                self.linear_solver.post_singular_warning()    # (check_singular=True: the reference warns from inside solve())
            if converged is not None and not (info.status == NonlinearOptimizerStatus.FAIL).any():
                cm = converged.cpu().numpy()
                info.status[cm] = NonlinearOptimizerStatus.CONVERGED
                info.converged_iter = conv_iter.cpu()
            info.status[info.status == NonlinearOptimizerStatus.START] = NonlinearOptimizerStatus.MAX_ITERATIONS
            info.converged_iter[torch.from_numpy(info.status == NonlinearOptimizerStatus.MAX_ITERATIONS)] = -1
            if err_hist is not None:
                info.err_history = err_hist.cpu()
            if state_hist is not None:
                info.state_history = state_hist.to_dict(packed)
            if track_best_solution:
                info.best_err = best_err
                info.best_iter = best_iter.cpu()
                info.best_solution = packed.solution_dict(best_state)
        return info

    def _needs_grad(self):
        return any(v.tensor.requires_grad for v in self.linear_solver.linearization.packed._tracked())

    def _implicit_last_step(self, packed, outer_grad: bool, kwargs):
        """X_new = X exp(delta), delta = H^-1 g(theta): H is built outside autograd (i.e. detached, as
        dense_linearization.py:61 does for this step), the graph runs through g only and its backward is one
        linear solve with the cached factor + the fused VJP kernel (theseus_amd/autograd.py)."""
        from .autograd import ImplicitStep
        step = self.params.step_size if kwargs.get("__keep_final_step_size__", False) else 1.0
        if getattr(packed, "group", None) == "BA":   # bundle adjustment: theseus_amd/ba.py (thx_ba_vjp)
            from .ba import ba_implicit_step
            with torch.set_grad_enabled(outer_grad):
                return ba_implicit_step(self, packed, float(step), kwargs)
        if packed.group == "Euclidean":   # generic objectives: theseus_amd/euclidean.py (torch forms g, cached-factor solve)
            with torch.set_grad_enabled(outer_grad):
                return packed.implicit_step(self, float(step), kwargs)
        if packed.group not in ("SE3", "SE2", "SO3"):
            raise NotImplementedError("HIP back end: backward_mode='implicit' is fused for SE3 / SE2 / SO3 pose graphs and "
                                      f"bundle adjustment (got {packed.group}); there is no autograd/CPU fallback.")
        with torch.set_grad_enabled(outer_grad):
            packed.flush_variables()
            packed.sync(force=True)  # re-pack the auxiliary tensors WITH their autograd history
            t = packed.tensors
            return ImplicitStep.apply(self, float(step), kwargs, t.meas, t.w_between, t.prior_target, t.w_prior,
                                      t.log_radius_between, t.log_radius_prior)


class GaussNewton(NonlinearLeastSquares):
    def compute_delta(self, **kwargs) -> torch.Tensor:
        return self.linear_solver.solve(check_info=False)


class LevenbergMarquardt(NonlinearLeastSquares):
    _MIN_DAMPING = 1.0e-7
    _MAX_DAMPING = 1.0e7

    def __init__(self, objective: Objective, *args, **kwargs):
        super().__init__(objective, *args, **kwargs)
        self._damping: Union[float, torch.Tensor] = 0.001
        self._reject: Optional[torch.Tensor] = None

    # levenberg_marquardt.py:90-110
    def reset(self, damping: float = 1e-3, adaptive_damping: bool = False, **kwargs) -> None:
        super().reset(**kwargs)
        packed = self.linear_solver.linearization.packed
        packed.sync()
        if adaptive_damping:
            self._damping = damping * torch.ones(packed.batch, device=packed.device,
                                                 dtype=self.objective.dtype)
            self._reject = torch.zeros(packed.batch, dtype=torch.uint8, device=packed.device)
        else:
            self._damping = damping

    def _may_reject(self, kwargs) -> bool:
        return bool(kwargs.get("adaptive_damping", False))

    # levenberg_marquardt.py:114-137
    def compute_delta(self, ellipsoidal_damping: bool = False, damping_eps: Optional[float] = None,
                      **kwargs) -> torch.Tensor:
        damping_eps = damping_eps if damping_eps is not None else 1e-8
        return self.linear_solver.solve(damping=self._damping, ellipsoidal_damping=ellipsoidal_damping,
                                        damping_eps=damping_eps, check_info=False)

    # levenberg_marquardt.py:139-201 (fused: thx_lm_accept)
    def _complete_step(self, delta, new_err, previous_err, step_size: float = 1.0, adaptive_damping: bool = False,
                       down_damping_ratio: float = 9.0, up_damping_ratio: float = 11.0,
                       damping_accept: float = 0.1, ellipsoidal_damping: bool = False, **kwargs):
        if not adaptive_damping:
            return None
        lin = self.linear_solver.linearization
        d = delta if step_size == 1.0 else delta * step_size
        lin.lm_accept(d, self._damping, previous_err, new_err, ellipsoidal_damping, damping_accept, down_damping_ratio,
                      up_damping_ratio, self._reject)
        return self._reject


class TrustRegion(NonlinearLeastSquares, abc.ABC):
    """theseus/optimizer/nonlinear/trust_region.py:35-151: per-problem trust-region radius, gain ratio against the quadratic
    model, shrink / expand / reject.  Everything stays on the device (no host decision)."""

    def __init__(self, objective: Objective, *args, **kwargs):
        super().__init__(objective, *args, **kwargs)
        self._trust_region: Optional[torch.Tensor] = None

    # trust_region.py:65-76
    def reset(self, trust_region_init: float = 0.5, **kwargs) -> None:
        super().reset(**kwargs)
        packed = self.linear_solver.linearization.packed
        packed.sync()
        self._trust_region = trust_region_init * torch.ones(packed.batch, 1, device=packed.device, dtype=self.objective.dtype)

    def _may_reject(self, kwargs) -> bool:
        return True

    @abc.abstractmethod
    def _compute_delta_impl(self) -> torch.Tensor:
        pass

    def compute_delta(self, **kwargs) -> torch.Tensor:
        return self._compute_delta_impl()

    @staticmethod
    def _squared_norm(tensor: torch.Tensor, keepdim: bool = True) -> torch.Tensor:
        return (tensor ** 2).sum(dim=1, keepdim=keepdim)

    # trust_region.py:91-105: m(delta) = err + delta . grad + |A delta|^2 / 2 with grad = -Atb
    def _predicted_error(self, previous_error: torch.Tensor, delta: torch.Tensor) -> torch.Tensor:
        lin = self.linear_solver.linearization
        Adelta = lin.Av(delta)
        grad = -lin.Atb.squeeze(2)
        return previous_error + (delta * grad).sum(dim=1) + 0.5 * TrustRegion._squared_norm(Adelta, keepdim=False)

    # trust_region.py:113-151
    def _complete_step(self, delta, new_err, previous_err, step_size: float = 1.0, accept_threshold: float = 0.0,
                       shrink_threshold: float = 0.25, expand_threshold: float = 0.75, shrink_ratio: float = 0.25,
                       expand_ratio: float = 2.0, min_trust_region: float = 1.0e-5, max_trust_region: float = 1.0e5,
                       **kwargs) -> Optional[torch.Tensor]:
        good_params = (0.0 < shrink_ratio <= 1.0) and (expand_ratio >= 1.0)
        good_params &= (shrink_threshold < expand_threshold) and (accept_threshold < shrink_threshold)
        if not good_params:
            raise ValueError("Invalid parameters for TrustRegionMethod. Values must satisfy <accept/shrink>_threshold < "
                             "expand_threshold, shrink_ratio in (0, 1], and expand_ratio > 1.0.")
        d = delta if step_size == 1.0 else delta * step_size
        pred_err = self._predicted_error(previous_err, d)
        rho = ((previous_err - new_err) / (previous_err - pred_err)).view(-1, 1)
        tr = self._trust_region
        tr = torch.where(rho < shrink_threshold, tr * shrink_ratio, tr)
        tr = torch.where(rho > expand_threshold, tr * expand_ratio, tr)
        self._trust_region = tr.clamp(min_trust_region, max_trust_region)
        return (rho < accept_threshold).view(-1)


class Dogleg(TrustRegion):
    """theseus/optimizer/nonlinear/dogleg.py:18-116 (Nocedal & Wright, pp. 73-77)."""
    EPS = 1e-7

    def _join_compute_delta(self):
        one = torch.ones((), dtype=torch.bool, device=self._trust_region.device)
        self.reducer.device_all(one)

    def _compute_delta_impl(self) -> torch.Tensor:
        sq = TrustRegion._squared_norm
        tr = self._trust_region
        tr2 = tr ** 2
        delta_gn = self.linear_solver.solve(check_info=False)
        # dogleg.py:55-58 returns the Gauss-Newton steps untouched when ALL of them (the whole batch, every shard) are inside
        # their regions -- a batch-global predicate, kept on the device and applied as a select
        all_inside = self.reducer.device_all((sq(delta_gn) < tr2).all())
        lin = self.linear_solver.linearization
        delta_sd = lin.Atb.squeeze(2)
        Asd2 = sq(lin.Av(delta_sd))
        g2 = sq(delta_sd)
        cauchy = g2 / (Asd2 + Dogleg.EPS)
        delta_c = delta_sd * cauchy
        c2 = g2 * cauchy ** 2
        inside = c2 <= tr2
        # steps beyond the region are truncated (dogleg.py:75-82) ...
        out = torch.where(inside, delta_c, delta_c * tr / (c2 + Dogleg.EPS).sqrt())
        # ... Cauchy steps inside it are extended towards the Gauss-Newton step up to the boundary (:84-101)
        diff = delta_gn - delta_c
        a = sq(diff)
        b = (2 * delta_c * diff).sum(dim=1, keepdim=True)
        c = c2 - tr2
        disc = (b ** 2 - 4 * a * c).clamp(Dogleg.EPS)
        tau = ((-b + disc.sqrt()) / (2 * a + Dogleg.EPS)).clamp(max=1.0)
        out = torch.where(inside, delta_c + tau * diff, out)
        return torch.where(all_inside, delta_gn, out)
