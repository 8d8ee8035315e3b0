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


@dataclass
class NonlinearOptimizerParams:
    abs_err_tolerance: float
    rel_err_tolerance: float
    max_iterations: int
    step_size: float


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
            linear_solver_cls = HipSchurSolver if "Point3" in kinds else HipCholeskySolver
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

    def _optimize_impl(self, track_best_solution: bool = False, track_err_history: bool = False,
                       track_state_history: bool = False, verbose: bool = False,
                       backward_mode: Union[str, BackwardMode] = BackwardMode.UNROLL,
                       end_iter_callback: Optional[Callable] = None, **kwargs) -> NonlinearOptimizerInfo:
        backward_mode = BackwardMode.resolve(backward_mode)
        outer_grad = torch.is_grad_enabled()
        if backward_mode in (BackwardMode.TRUNCATED, BackwardMode.DLM):
            raise NotImplementedError(f"backward_mode={backward_mode.name} is not supported by the HIP back end "
                                      "(supported: 'implicit'; 'unroll' without gradients).")
        if backward_mode == BackwardMode.UNROLL and outer_grad and self._needs_grad():
            raise NotImplementedError(
                "Differentiating through the unrolled iterations (backward_mode='unroll') is not supported by the "
                "HIP back end: the kernels are outside autograd.  Use backward_mode='implicit' (one backward linear "
                "solve with the cached factor), or call under torch.no_grad().")
        if track_state_history:
            raise NotImplementedError("track_state_history is not supported by the HIP back end (the iterates live in two "
                                      "recycled device buffers); use end_iter_callback to copy the states you need.")
        implicit = backward_mode == BackwardMode.IMPLICIT
        lin: HipLinearization = self.linear_solver.linearization
        packed = lin.packed
        with torch.no_grad():
            packed.sync(deep=True)   # once per optimize(): also catches in-place edits of the variables' tensors
        self.reset(**kwargs, backward_mode=backward_mode)
        _, loop_iters = self._split_backward_iters(backward_mode=backward_mode, **kwargs) if implicit else (0, self.params.max_iterations)
        with torch.no_grad():
            packed.sync()
            packed.privatize_state()   # never recycle the buffer the previous optimize() handed to the user
            B = packed.batch
            dev, dt = packed.device, self.objective.dtype
            p = self.params
            last_err = packed.error_metric()
            err_hist = None
            if track_err_history:
                err_hist = torch.full((B, p.max_iterations + 1), float("inf"), dtype=dt, device=dev)
                err_hist[:, 0] = last_err
            info = NonlinearOptimizerInfo(
                best_solution=None, status=np.array([NonlinearOptimizerStatus.START] * B),
                converged_iter=torch.full((B,), -1, dtype=torch.long), best_iter=torch.zeros(B, dtype=torch.long),
                err_history=err_hist, last_err=last_err, best_err=last_err.clone())
            if track_best_solution:
                best_state = packed.clone_state()
                best_err = last_err.clone()
                best_iter = torch.zeros(B, dtype=torch.long, device=dev)
            if verbose:
                print(f"Nonlinear optimizer. Iteration: 0. Error: {last_err.mean().item()}")

            need_conv = p.abs_err_tolerance > 0 or p.rel_err_tolerance > 0
            converged = None          # (B,) bool on device, None == nobody
            conv_iter = torch.full((B,), -1, dtype=torch.long, device=dev)
            spare = packed.alloc_state()
            err_new = torch.empty(B, dtype=dt, device=dev)
            it, all_reject_attempts = 0, 0
            # ---- sync-free iterations.  Without adaptive damping, convergence tests, callbacks or a
            #      sharded batch the only host decision of an iteration is "did a linear solve fail?"
            #      (nonlinear_least_squares.py:138-152: FAIL status, variables keep their values).  That flag stays on
            #      the device (OR-ed over the shards by a stream-ordered all-reduce when the batch is sharded): once
            #      raised it freezes every later update through the retraction mask, and it is read ONCE after the loop
            #      -- the host queues the iterations back to back and the GPU never drains.  (Local AND sharded batches:
            #      DistBatchReducer is a LocalBatchReducer whose device_any() all-reduces.) ----
            lazy = (isinstance(self.reducer, LocalBatchReducer) and not need_conv and end_iter_callback is None
                    and not verbose and not kwargs.get("adaptive_damping", False))
            failed = first_fail = None
            if lazy:
                failed = torch.zeros((), dtype=torch.bool, device=dev)
                first_fail = torch.full((), -1, dtype=torch.long, device=dev)
            raised = None
            while lazy and it < loop_iters:
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
                    delta = torch.zeros(B, lin.num_cols, dtype=dt, device=dev)
                    local_fail = torch.ones((), dtype=torch.bool, device=dev)
                now = self.reducer.device_any(local_fail)
                first_fail = torch.where(now & ~failed, torch.full_like(first_fail, it), first_fail)
                failed = failed | now
                packed.retract(delta, p.step_size, failed.to(torch.uint8).expand(B).contiguous(), spare)
                packed.error_metric(state=spare, out=err_new)
                err = torch.where(failed, last_err, err_new)
                spare = packed.swap_state(spare)
                if err_hist is not None:
                    err_hist[:, it + 1] = torch.where(failed, torch.full_like(err, float("inf")), err)
                if track_best_solution:
                    better = (err < best_err) & ~failed
                    packed.copy_where(better, packed.state, best_state)
                    best_err = torch.where(better, err, best_err)
                    best_iter = torch.where(better, torch.full_like(best_iter, it), best_iter)  # nonlinear_optimizer.py:202
                last_err = err
                info.last_err = err
                it += 1
                info.iters_done = it
            if lazy and bool(failed):  # the one host sync of the loop
                try:
                    if raised is not None:
                        raise raised
                    self.linear_solver.check_info()
                    raise RuntimeError("the linear solve failed on another shard of the batch")
                except RuntimeError as run_err:
                    warnings.warn(f"There was an error while running the linear optimizer. "
                                  f"Original error message: {run_err}.", RuntimeWarning)
                info.status[:] = NonlinearOptimizerStatus.FAIL
                info.iters_done = int(first_fail)
            while not lazy and it < loop_iters:
                lin.linearize()
                try:
                    delta = self.compute_delta(**kwargs)
                except RuntimeError as run_err:
                    msg = f"There was an error while running the linear optimizer. Original error message: {run_err}."
                    warnings.warn(msg, RuntimeWarning)
                    info.status[:] = NonlinearOptimizerStatus.FAIL
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
                        warnings.warn(f"There was an error while running the linear optimizer. "
                                      f"Original error message: {run_err}.", RuntimeWarning)
                    info.status[:] = NonlinearOptimizerStatus.FAIL
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
                        break
                last_err = err
                info.last_err = err
                if end_iter_callback is not None:
                    packed.flush_variables()
                    end_iter_callback(self, info, delta, it)
                it += 1
                info.iters_done = it

            # ---- BackwardMode.IMPLICIT: the last step is an undamped Gauss-Newton step with the Hessian detached,
            #      executed under the caller's grad mode (nonlinear_least_squares.py:121-135,265-292) ----
            if implicit and not (info.status == NonlinearOptimizerStatus.FAIL).any():
                X_new, delta = self._implicit_last_step(packed, outer_grad, kwargs)
                err = packed.error_metric(state=X_new.detach())
                packed.swap_state(X_new)
                if err_hist is not None:
                    err_hist[:, it + 1] = err
                if track_best_solution:
                    better = err < best_err
                    packed.copy_where(better, X_new.detach(), best_state)
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
            if converged is not None:
                cm = converged.cpu().numpy()
                info.status[cm] = NonlinearOptimizerStatus.CONVERGED
                info.converged_iter = conv_iter.cpu()
            info.status[info.status == NonlinearOptimizerStatus.START] = NonlinearOptimizerStatus.MAX_ITERATIONS
            info.converged_iter[torch.from_numpy(info.status == NonlinearOptimizerStatus.MAX_ITERATIONS)] = -1
            if err_hist is not None:
                info.err_history = err_hist.cpu()
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
        if packed.group not in ("SE3", "SE2"):
            raise NotImplementedError("HIP back end: backward_mode='implicit' is fused for SE3 / SE2 pose graphs "
                                      f"(got {packed.group}); there is no autograd/CPU fallback.")
        step = self.params.step_size if kwargs.get("__keep_final_step_size__", False) else 1.0
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
