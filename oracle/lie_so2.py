"""Oracle: SO2 as a group of its own (TEST INFRASTRUCTURE, see oracle/__init__.py).

theseus/geometry/so2.py: tensor (...,2) = [cos, sin], tangent theta (1), right perturbations.  exp_map :167-186 with
update_from_angle :96-100 (cos / sin of the angle, Jacobian 1), _log_map_impl :206-223 (atan2(sin, cos), Jacobian 1),
_adjoint_impl :116-117 (the scalar 1), _compose_impl :225-231 (angle-addition formulas), _inverse_impl :233-235 ((cos, -sin)).
No Taylor switches on this path and no re-normalisation (constructors pass tensors through).  Plain torch ops: autograd
differentiates the closed forms, exactly like the reference (so2.py has no custom backward).
"""
import torch

DOF, NS = 1, 2


def so2_exp(theta):
    """(...,1) -> (...,2)."""
    t = theta[..., 0]
    return torch.stack([t.cos(), t.sin()], -1)


def so2_exp_jexp(theta):
    return so2_exp(theta), torch.ones(*theta.shape[:-1], 1, 1, dtype=theta.dtype)


def so2_log_jlog(X):
    theta = torch.atan2(X[..., 1], X[..., 0]).unsqueeze(-1)
    return theta, torch.ones(*X.shape[:-1], 1, 1, dtype=X.dtype)


def so2_log(X):
    return so2_log_jlog(X)[0]


def so2_compose(A, B):
    c1, s1, c2, s2 = A[..., 0], A[..., 1], B[..., 0], B[..., 1]
    return torch.stack([c1 * c2 - s1 * s2, s1 * c2 + c1 * s2], -1)


def so2_inverse(X):
    return torch.stack([X[..., 0], -X[..., 1]], -1)


def so2_adjoint(X):
    return torch.ones(*X.shape[:-1], 1, 1, dtype=X.dtype)


def so2_retract(X, delta):
    """lie_group.py:197-198: X exp(delta)."""
    return so2_compose(X, so2_exp(delta))
