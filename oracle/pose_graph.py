"""Oracle: dense GN/LM on SE3 pose graphs, restated from the reference (TEST INFRASTRUCTURE).

Follows, on packed tensors instead of per-variable Python objects:
  Between.jacobians                 theseus/embodied/measurements/between.py:38-45
  Local/Difference.jacobians        theseus/embodied/misc/local_cost_fn.py:39-61, theseus/geometry/lie_group.py:180-195
  Diagonal/ScaleCostWeight          theseus/core/cost_weight.py:81-90,125-136
  DenseLinearization                theseus/optimizer/dense_linearization.py:29-62
  DenseSolver._apply_damping        theseus/optimizer/linear/dense_solver.py:38-64
  CholeskyDenseSolver._solve_sytem  theseus/optimizer/linear/dense_solver.py:159-161
  RobustCostFunction (incl. flatten_dims) / Welsch,Huber theseus/core/robust_cost_function.py:87-135, theseus/core/robust_loss.py:13-52
  retract / error_metric            theseus/core/objective.py:37-38,562-641,873-914, theseus/core/variable.py:65-69
  LM loop                           theseus/optimizer/nonlinear/nonlinear_least_squares.py:100-215,338-365
                                    theseus/optimizer/nonlinear/levenberg_marquardt.py:114-201
                                    theseus/optimizer/nonlinear/nonlinear_optimizer.py:110-119
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch

from . import lie, lie_se2, lie_so2, lie_so3

MIN_DAMPING, MAX_DAMPING = 1.0e-7, 1.0e7  # levenberg_marquardt.py:52-53


class _SE3:
    name, dof = "SE3", 6
    compose, inverse, adjoint, retract = (staticmethod(lie.se3_compose), staticmethod(lie.se3_inverse),
                                          staticmethod(lie.se3_adjoint), staticmethod(lie.se3_retract))
    log_jlog = staticmethod(lie.se3_log_jlog_autograd)


class _SE2:
    name, dof = "SE2", 3
    compose, inverse, adjoint, retract = (staticmethod(lie_se2.se2_compose), staticmethod(lie_se2.se2_inverse),
                                          staticmethod(lie_se2.se2_adjoint), staticmethod(lie_se2.se2_retract))
    log_jlog = staticmethod(lie_se2.se2_log_jlog)


class _SO3:
    name, dof = "SO3", 3
    compose, inverse, adjoint, retract = (staticmethod(lie_so3.so3_compose), staticmethod(lie_so3.so3_inverse),
                                          staticmethod(lie_so3.so3_adjoint), staticmethod(lie_so3.so3_retract))
    log_jlog = staticmethod(lie_so3.so3_log_jlog_autograd)


class _SO2:
    name, dof = "SO2", 1
    compose, inverse, adjoint, retract = (staticmethod(lie_so2.so2_compose), staticmethod(lie_so2.so2_inverse),
                                          staticmethod(lie_so2.so2_adjoint), staticmethod(lie_so2.so2_retract))
    log_jlog = staticmethod(lie_so2.so2_log_jlog)


GROUPS = {"SE3": _SE3, "SE2": _SE2, "SO3": _SO3, "SO2": _SO2}


@dataclass
class PGProblem:
    """Packed pose-graph problem.  Column order = pose index order (VariableOrdering default =
    insertion order, theseus/optimizer/variable_ordering.py:19-27); row order = ``cost_order``."""

    num_poses: int
    edges: torch.Tensor          # (E,2) long, (i,j): Between(v0=pose i, v1=pose j)
    meas: torch.Tensor           # (B,E,3,4)
    w_between: torch.Tensor      # (1|B, E, 6)  sqrt-information diagonal
    prior_idx: torch.Tensor      # (K,) long
    prior_target: torch.Tensor   # (1|B, K, 3, 4)
    w_prior: torch.Tensor        # (1|B, K, 6)
    cost_order: Optional[List[Tuple[str, int]]] = None  # [('between',k)|('prior',k)] add order
    group: str = "SE3"           # "SE3": tensors (...,3,4), dof 6; "SE2": tensors (...,4) = [x,y,cos,sin], dof 3
    # RobustCostFunction wrappers (robust_cost_function.py): loss spec per cost role -- "welsch" | "huber" | None, with
    # "+flatten" appended for flatten_dims=True, or a LIST of such specs, one per cost of the role -- and log_loss_radius
    # broadcastable to (B, E|K, 1) (the entries of plain costs are not read)
    robust_between: Optional[object] = None
    log_radius_between: Optional[torch.Tensor] = None
    robust_prior: Optional[object] = None
    log_radius_prior: Optional[torch.Tensor] = None

    def __post_init__(self):
        if self.cost_order is None:
            self.cost_order = [("between", k) for k in range(self.edges.shape[0])] + [
                ("prior", k) for k in range(self.prior_idx.shape[0])
            ]

    @property
    def G(self):
        return GROUPS[self.group]

    @property
    def dof(self):
        return self.G.dof

    @property
    def n(self):
        return self.dof * self.num_poses

    @property
    def m(self):
        return self.dof * len(self.cost_order)

    def row_starts(self):
        rs = {}
        for r, c in enumerate(self.cost_order):
            rs[c] = self.dof * r
        return rs


def between_jac_err(v0, v1, meas, w, G=_SE3):
    """between.py:38-45 + cost_weight.py:125-136.  Group tensors x3, w (...,dof)."""
    D = G.compose(G.inverse(v0), v1)
    E = G.compose(G.inverse(meas), D)
    e, Jlog = G.log_jlog(E)
    J0 = -Jlog @ G.adjoint(G.inverse(D))
    J1 = Jlog
    return J0 * w[..., :, None], J1 * w[..., :, None], e * w


def local_jac_err(target, var, w, G=_SE3):
    """local_cost_fn.py:58-61 -> lie_group.py:180-195: e = log(target^-1 var), J = Jlog."""
    D = G.compose(G.inverse(target), var)
    e, Jlog = G.log_jlog(D)
    return Jlog * w[..., :, None], e * w


_LOSS_EPS = 1e-20    # robust_loss.py:10
_ROBUST_EPS = 1e-20  # robust_cost_function.py:52


def loss_evaluate(kind, x, log_radius):
    """robust_loss.py:13-16,33-36,43-47: rho(x), x = squared norm of the weighted error."""
    r = log_radius.exp()
    if kind == "welsch":
        return r - r * torch.exp(-x / (r + _LOSS_EPS))
    if kind == "huber":
        return torch.where(x > r, 2 * torch.sqrt(r * torch.maximum(x, r) + _LOSS_EPS) - r, x)
    if kind == "hinge":          # robust_loss.py:56-58
        return torch.where(x > r, torch.sqrt(x) - torch.sqrt(r), torch.full_like(x, _LOSS_EPS))
    if kind == "gm":             # robust_loss.py:96-107, Geman-McClure; r = mu * radius (log_radius carries log(mu * radius))
        return r * x / (r + x + _LOSS_EPS)
    raise ValueError(kind)


def loss_linearize(kind, x, log_radius):
    """robust_loss.py:18-20,38-40,49-52: rho'(x)."""
    r = log_radius.exp()
    if kind == "welsch":
        return torch.exp(-x / (r + _LOSS_EPS))
    if kind == "huber":
        return torch.sqrt(r / torch.maximum(x, r) + _LOSS_EPS)
    if kind == "hinge":          # robust_loss.py:60-62
        return torch.where(x > r, 1.0 / (2 * torch.sqrt(x) + _LOSS_EPS), torch.zeros_like(x))
    if kind == "gm":             # robust_loss.py:109-113
        return r ** 2 / ((r + x) ** 2 + _LOSS_EPS)
    raise ValueError(kind)


def _per_cost(kind, count):
    """A loss spec -- None | "welsch" | "huber" | "hinge" | "gm" (Geman-McClure: the radius entry is log(mu * radius)) [+ "+flatten" for flatten_dims=True], or one such entry per cost of the role
    (plain, Welsch, Huber and flattened costs mixed) -- as (count, 1) tensors: kind index 0/1/2/3/8, flatten flag."""
    specs = [kind] * count if kind is None or isinstance(kind, str) else list(kind)
    if len(specs) != count:
        raise ValueError("one loss spec per cost")
    k = torch.tensor([0 if s is None else {"welsch": 1, "huber": 2, "hinge": 3, "gm": 8}[s.split("+")[0]] for s in specs]).view(count, 1)
    f = torch.tensor([s is not None and s.endswith("+flatten") for s in specs]).view(count, 1)
    return k, f


def _simple(kind):
    return kind is None or (isinstance(kind, str) and "+" not in kind)


def _both(fn, k, x, log_radius):
    return torch.where(k == 1, fn("welsch", x, log_radius),
                       torch.where(k == 3, fn("hinge", x, log_radius),
                                   torch.where(k == 8, fn("gm", x, log_radius), fn("huber", x, log_radius))))


def robust_rescale(jacs, e, kind, log_radius):
    """robust_cost_function.py:115-135: J, e <- sqrt(rho'(|e|^2) + eps) * (J, e); with flatten_dims (:118-133) every row r is
    its own term: row r of (J, e) scaled by sqrt(rho'(e_r^2) + eps)."""
    if kind is None:
        return jacs, e
    if _simple(kind):
        sqn = (e**2).sum(-1, keepdim=True)
        rs = (loss_linearize(kind, sqn, log_radius) + _ROBUST_EPS).sqrt()
        return [rs.unsqueeze(-1) * J for J in jacs], rs * e
    k, f = _per_cost(kind, e.shape[1])
    x = torch.where(f, e**2, (e**2).sum(-1, keepdim=True).expand_as(e))
    rs = torch.where(k == 0, torch.ones_like(x), (_both(loss_linearize, k, x, log_radius) + _ROBUST_EPS).sqrt())
    return [rs.unsqueeze(-1) * J for J in jacs], rs * e


def robust_weighted_error(e, kind, log_radius):
    """robust_cost_function.py:87-106: ones * sqrt(rho(|e|^2) / dim + eps), so that |h|^2 = rho (+ dim eps); with flatten_dims
    (:89-96) sqrt(rho(e_r^2) + eps) per row."""
    if kind is None:
        return e
    if _simple(kind):
        sqn = (e**2).sum(-1, keepdim=True)
        return torch.ones_like(e) * (loss_evaluate(kind, sqn, log_radius) / e.shape[-1] + _ROBUST_EPS).sqrt()
    k, f = _per_cost(kind, e.shape[1])
    x = torch.where(f, e**2, (e**2).sum(-1, keepdim=True).expand_as(e))
    rho = _both(loss_evaluate, k, x, log_radius)
    h = torch.where(f, (rho + _ROBUST_EPS).sqrt(), (rho / e.shape[-1] + _ROBUST_EPS).sqrt())
    return torch.where(k == 0, e, h)


def cost_terms(p: PGProblem, poses):
    """weighted_jacobians_error of every cost (robust ones rescaled): J0, J1, eb, Jp, ep."""
    i, j = p.edges[:, 0], p.edges[:, 1]
    J0, J1, eb = between_jac_err(poses[:, i], poses[:, j], p.meas, p.w_between, p.G)
    (J0, J1), eb = robust_rescale([J0, J1], eb, p.robust_between, p.log_radius_between)
    Jp, ep = local_jac_err(p.prior_target, poses[:, p.prior_idx], p.w_prior, p.G)
    (Jp,), ep = robust_rescale([Jp], ep, p.robust_prior, p.log_radius_prior)
    return J0, J1, eb, Jp, ep


def weighted_errors(p: PGProblem, poses):
    """weighted_error of all costs: (e_between (B,E,6), e_prior (B,K,6)); robust costs return their
    sqrt(rho/dim) vector (robust_cost_function.py:87-106)."""
    i, j = p.edges[:, 0], p.edges[:, 1]
    _, _, eb = between_jac_err(poses[:, i], poses[:, j], p.meas, p.w_between, p.G)
    _, ep = local_jac_err(p.prior_target, poses[:, p.prior_idx], p.w_prior, p.G)
    return (robust_weighted_error(eb, p.robust_between, p.log_radius_between),
            robust_weighted_error(ep, p.robust_prior, p.log_radius_prior))


def error_vector(p: PGProblem, poses):
    """objective.py:562-613: concatenated weighted error (B,m) in cost add order."""
    eb, ep = weighted_errors(p, poses)
    cols = [eb[:, k] if kind == "between" else ep[:, k] for kind, k in p.cost_order]
    return torch.cat(cols, dim=1)


def error_metric(p: PGProblem, poses):
    """objective.py:37-38,615-641: 0.5 * sum(e^2) per problem."""
    eb, ep = weighted_errors(p, poses)
    return 0.5 * ((eb**2).sum((1, 2)) + (ep**2).sum((1, 2)))


def dense_linearize(p: PGProblem, poses):
    """dense_linearization.py:29-56: dense A (B,m,n), b = -err (B,m)."""
    B = poses.shape[0]
    i, j = p.edges[:, 0], p.edges[:, 1]
    J0, J1, eb, Jp, ep = cost_terms(p, poses)
    A = torch.zeros(B, p.m, p.n, dtype=poses.dtype)
    b = torch.zeros(B, p.m, dtype=poses.dtype)
    d = p.dof
    for r, (kind, k) in enumerate(p.cost_order):
        rs = d * r
        if kind == "between":
            ci, cj = d * int(i[k]), d * int(j[k])
            A[:, rs:rs + d, ci:ci + d] = J0[:, k]
            A[:, rs:rs + d, cj:cj + d] = J1[:, k]
            b[:, rs:rs + d] = -eb[:, k]
        else:
            c = d * int(p.prior_idx[k])
            A[:, rs:rs + d, c:c + d] = Jp[:, k]
            b[:, rs:rs + d] = -ep[:, k]
    return A, b


def hessian(A, b):
    """dense_linearization.py:58-62: AtA = A^T A, Atb = A^T b (B,n,1)."""
    At = A.transpose(1, 2)
    return At.bmm(A), At.bmm(b.unsqueeze(2))


def apply_damping(AtA, damping, ellipsoidal, eps):
    """dense_solver.py:38-64 (out of place)."""
    damping = torch.as_tensor(damping).to(dtype=AtA.dtype)
    n = AtA.shape[1]
    if ellipsoidal:
        damping = damping.view(-1, 1)
        D = torch.diag_embed(damping * AtA.diagonal(dim1=1, dim2=2) + eps)
    else:
        damping = damping.view(-1, 1, 1)
        D = damping * torch.eye(n, dtype=AtA.dtype).unsqueeze(0)
    return AtA + D


def cholesky_solve(Atb, AtA):
    """dense_solver.py:159-161."""
    L = torch.linalg.cholesky(AtA)
    return torch.cholesky_solve(Atb, L).squeeze(2)


def solve(AtA, Atb, damping=None, ellipsoidal=False, eps=1e-8):
    """dense_solver.py:66-79,84-123."""
    if damping is not None:
        AtA = apply_damping(AtA, damping, ellipsoidal, eps)
    return cholesky_solve(Atb, AtA)


def retract(poses, delta, ignore_mask=None, G=None):
    """objective.py:873-914 / vectorizer.py:410-469 / variable.py:65-69: X.exp(delta), masked rows keep X."""
    B, P = poses.shape[:2]
    if G is None:
        G = (_SO2 if poses.shape[-1] == 2 else _SE2) if poses.ndim == 3 else (_SO3 if poses.shape[-1] == 3 else _SE3)
    new = G.retract(poses, delta.view(B, P, G.dof))
    if ignore_mask is not None:
        new = torch.where(ignore_mask.view([B] + [1] * (poses.ndim - 1)), poses, new)
    return new


@dataclass
class LMInfo:
    err_history: List[torch.Tensor] = field(default_factory=list)  # [(B,)], index 0 = initial
    deltas: List[torch.Tensor] = field(default_factory=list)
    AtA: List[torch.Tensor] = field(default_factory=list)
    Atb: List[torch.Tensor] = field(default_factory=list)
    dampings: List[torch.Tensor] = field(default_factory=list)
    rho: List[torch.Tensor] = field(default_factory=list)            # adaptive LM: gain ratio per step
    actual_reduction: List[torch.Tensor] = field(default_factory=list)
    prev_err: List[torch.Tensor] = field(default_factory=list)
    trust_regions: List[torch.Tensor] = field(default_factory=list)  # Dogleg: radius after every counted iteration
    iters_done: int = 0
    converged_iter: Optional[torch.Tensor] = None
    last_err: Optional[torch.Tensor] = None


def check_convergence(err, last_err, abs_tol, rel_tol):
    """nonlinear_optimizer.py:110-119."""
    if err.abs().mean() < abs_tol:
        return torch.ones_like(err).bool()
    change = last_err - err
    return (change.abs() < abs_tol) | ((change / last_err).abs() < rel_tol)


def _ops(p):
    """(dense_linearize, error_metric, retract, keep_where) of a problem: module functions for pose graphs, the
    problem's own methods for other problem kinds (oracle/ba.py: state = (cameras, points))."""
    if hasattr(p, "oracle_ops"):
        return p.oracle_ops()

    def keep_where(mask, old, new):
        return torch.where(mask.view([old.shape[0]] + [1] * (old.ndim - 1)), old, new)
    return (lambda x: dense_linearize(p, x), lambda x: error_metric(p, x),
            lambda x, d, m: retract(x, d, ignore_mask=m), keep_where)


def lm_optimize(p: PGProblem, poses, max_iterations=20, step_size=1.0, damping=1e-3,
                adaptive_damping=False, ellipsoidal_damping=False, damping_eps=1e-8,
                abs_err_tolerance=1e-10, rel_err_tolerance=1e-8,
                down_damping_ratio=9.0, up_damping_ratio=11.0, damping_accept=0.1,
                gauss_newton=False, keep_taps=False, dogleg=False, trust_region_init=0.5, accept_threshold=0.0,
                shrink_threshold=0.25, expand_threshold=0.75, shrink_ratio=0.25, expand_ratio=2.0,
                min_trust_region=1.0e-5, max_trust_region=1.0e5):
    """One ``optimize()`` of LevenbergMarquardt (GaussNewton if ``gauss_newton``, Dogleg if ``dogleg``) under no_grad.

    nonlinear_least_squares.py:100-215 (loop), :338-365 (_step), levenberg_marquardt.py:114-201, dogleg.py:52-116,
    trust_region.py:65-151, SURVEY Appendix B.  Returns (final poses, LMInfo).
    """
    _linearize, _error, _retract, _keep = _ops(p)
    first = poses[0] if isinstance(poses, (tuple, list)) else poses
    B = first.shape[0]
    dtype = first.dtype
    info = LMInfo()
    lam = damping * torch.ones(B, dtype=dtype) if adaptive_damping else damping
    trust_region = trust_region_init * torch.ones(B, 1, dtype=dtype)   # trust_region.py:65-76
    last_err = _error(poses)
    info.err_history.append(last_err.clone())
    converged = torch.zeros(B, dtype=torch.bool)
    info.converged_iter = torch.full((B,), -1, dtype=torch.long)
    it, all_reject_attempts = 0, 0
    while it < max_iterations:
        A, b = _linearize(poses)
        AtA, Atb = hessian(A, b)
        if dogleg:
            delta = dogleg_step(A, AtA, Atb, trust_region)
        elif gauss_newton:
            delta = solve(AtA, Atb)
        else:
            delta = solve(AtA, Atb, lam, ellipsoidal_damping, damping_eps)
        if keep_taps:
            info.AtA.append(AtA)
            info.Atb.append(Atb)
            info.deltas.append(delta)
            info.dampings.append(torch.as_tensor(lam).clone())
        d = delta * step_size
        new_poses = _retract(poses, d, converged)
        err = _error(new_poses)
        reject = None
        if adaptive_damping and not gauss_newton:
            dmp = lam.view(-1, 1)
            if ellipsoidal_damping:
                dmp = AtA.diagonal(dim1=1, dim2=2) * dmp  # linearization.diagonal_scaling
            den = (d * (dmp * d + Atb.squeeze(2))).sum(dim=1) / 2
            rho = (last_err - err) / den
            reject = rho <= damping_accept
            if keep_taps:
                info.rho.append(rho.clone())
                info.actual_reduction.append((last_err - err).clone())
                info.prev_err.append(last_err.clone())
            lam = torch.where(reject, lam * up_damping_ratio, lam / down_damping_ratio)
            lam = lam.clamp(MIN_DAMPING, MAX_DAMPING)
        if dogleg:
            # trust_region.py:91-151: rho against the quadratic model m(d) = err + d.grad + |A d|^2 / 2, grad = -Atb
            Ad = A.bmm(d.unsqueeze(2)).squeeze(2)
            pred = last_err + (d * -Atb.squeeze(2)).sum(dim=1) + 0.5 * (Ad ** 2).sum(dim=1)
            rho = ((last_err - err) / (last_err - pred)).view(-1, 1)
            trust_region = torch.where(rho < shrink_threshold, trust_region * shrink_ratio, trust_region)
            trust_region = torch.where(rho > expand_threshold, trust_region * expand_ratio, trust_region)
            trust_region = trust_region.clamp(min_trust_region, max_trust_region)
            reject = (rho < accept_threshold).view(-1)
            if keep_taps:
                info.rho.append(rho.view(-1).clone())
        if reject is not None and bool(reject.all()):
            all_reject_attempts += 1
            if all_reject_attempts < 3:  # nonlinear_optimizer.py:88 _MAX_ALL_REJECT_ATTEMPTS
                continue
            err = last_err
        else:
            if reject is not None:
                poses = _keep(reject, poses, new_poses)
                if bool(reject.any()):
                    err = _error(poses)
            else:
                poses = new_poses
        all_reject_attempts = 0
        info.err_history.append(err.clone())
        converged = check_convergence(err, last_err, abs_err_tolerance, rel_err_tolerance)
        info.converged_iter[converged & (info.converged_iter < 0)] = it + 1
        if bool(converged.all()):
            break  # nonlinear_least_squares.py:202-203 (breaks before counting the iteration)
        last_err = err
        if dogleg and keep_taps:   # (where the reference's end_iter_callback sees optimizer._trust_region)
            info.trust_regions.append(trust_region.view(-1).clone())
        it += 1
        info.iters_done = it
    info.last_err = info.err_history[-1]
    return poses, info


def dogleg_step(A, AtA, Atb, trust_region, eps=1e-7):
    """Dogleg._compute_delta_impl (dogleg.py:52-116; Nocedal & Wright pp. 73-77): the Gauss-Newton step where it lies inside
    the trust region of EVERY problem of the batch, otherwise per problem the Cauchy step (truncated to the region) extended
    towards the Gauss-Newton step up to the boundary.  ``trust_region``: (B, 1)."""
    sq = lambda t: (t ** 2).sum(dim=1, keepdim=True)  # noqa: E731   (TrustRegion._squared_norm)
    tr2 = trust_region ** 2
    delta_gn = solve(AtA, Atb)
    if bool((sq(delta_gn) < tr2).all()):
        return delta_gn
    delta_sd = Atb.squeeze(2)
    Asd2 = sq(A.bmm(delta_sd.unsqueeze(2)).squeeze(2))
    g2 = sq(delta_sd)
    cauchy = g2 / (Asd2 + eps)
    delta_c = delta_sd * cauchy
    c2 = g2 * cauchy ** 2
    inside = c2 <= tr2
    out = torch.where(inside, delta_c, delta_c * trust_region / (c2 + eps).sqrt())
    diff = delta_gn - delta_c
    a = sq(diff)
    b = (2 * delta_c * diff).sum(dim=1, keepdim=True)
    c = c2 - tr2
    disc = (b ** 2 - 4 * a * c).clamp(eps)
    tau = ((-b + disc.sqrt()) / (2 * a + eps)).clamp(max=1.0)
    return torch.where(inside, delta_c + tau * diff, out)


def implicit_final_step(p: PGProblem, poses, step_size=1.0, fallback_damping=None, ellipsoidal_damping=False,
                        damping_eps=1e-8):
    """The grad-enabled last step of BackwardMode.IMPLICIT (nonlinear_least_squares.py:121-135,265-292): undamped
    Gauss-Newton with the Hessian detached (dense_linearization.py:61); the autograd graph runs through
    Atb = A^T b only.  ``poses`` are the (detached) iterates of the no-grad loop; gradients flow to whatever in
    ``p`` requires grad (measurements, weights, prior targets).  If the undamped factorisation fails the reference falls
    back to the optimizer's regular (damped) step (nonlinear_least_squares.py:130-135): ``fallback_damping``, the LM damping
    (None: the error propagates, as with ``__strict_implicit_final_gn__``)."""
    A, b = dense_linearize(p, poses.detach())
    AtA, Atb = hessian(A, b)
    try:
        delta = solve(AtA.detach(), Atb)
    except RuntimeError:
        if fallback_damping is None:
            raise
        delta = solve(AtA.detach(), Atb, fallback_damping, ellipsoidal_damping, damping_eps)
    return retract(poses.detach(), delta * step_size), delta
