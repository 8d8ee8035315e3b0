"""Oracle: SO3 as a group of its own (TEST INFRASTRUCTURE, see oracle/__init__.py).

theseus/geometry/so3.py:99-186 wraps torchlie.functional.SO3 (torchlie/torchlie/functional/so3_impl.py): exp :220-261,
Jexp :270-320, log :390-433, Jlog :442-479, adjoint = R (:519-521), inverse = R^T (:556-558), compose = R0 R1 (:659-661).
Tensor (...,3,3); tangent w (3); right perturbations.  The closed forms themselves are the ones oracle/lie.py restates for
SE3's rotation part.
"""
import torch

from . import lie

EPS = lie.EPS


def so3_exp(w):
    return lie.so3_exp(w)


def so3_exp_jexp(w):
    """so3_impl.py:270-320: J = A I - hat(B w) + C w w^T, C = 0 below near_zero."""
    R, c = lie.so3_exp_helper(w)
    C = torch.where(c["nz"], torch.zeros_like(c["theta"]), (c["theta"] - c["sine"]) / (c["theta_nz"] * c["theta2_nz"]))
    J = C[..., None, None] * lie._outer(w, w) + c["A"][..., None, None] * torch.eye(3, dtype=w.dtype) - lie._hat(c["B"][..., None] * w)
    return R, J


def so3_log_jlog(R):
    w, c = lie.so3_log_helper(R)
    J, _ = lie.so3_jlog_helper(w, c["theta"], c["sine"], c["cosine"])
    return w, J


class _LogPassthrough(torch.autograd.Function):
    """What autograd sees when the reference asks ``SO3.log(R, jacobians=[...])`` (torchlie/functional/lie_group.py:60-84,
    148-155): the VALUE is the formula's, the BACKWARD is ``_log_backward`` (so3_impl.py:489-496):
    grad_R = R @ lift(J^T g / 2) -- the tangent projection, not the derivative of the closed form; the Jacobian itself
    keeps its plain autograd graph."""

    @staticmethod
    def forward(ctx, R, w, J):
        ctx.save_for_backward(R, J)
        return w.clone()

    @staticmethod
    def backward(ctx, g):
        R, J = ctx.saved_tensors
        h = 0.5 * (J.transpose(-1, -2) @ g.unsqueeze(-1)).squeeze(-1)
        return R @ lie._hat(h), None, None


def so3_log_jlog_autograd(R):
    """so3_log_jlog with the reference's autograd semantics (see _LogPassthrough); same values."""
    w, J = so3_log_jlog(R)
    return _LogPassthrough.apply(R, w, J), J


def so3_adjoint(R):
    return R.clone()


def so3_inverse(R):
    return R.transpose(-1, -2)


def so3_compose(A, B):
    return A @ B


def so3_retract(R, delta):
    return R @ lie.so3_exp(delta)
