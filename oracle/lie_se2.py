"""Oracle: SE2 arithmetic restated from theseus (TEST INFRASTRUCTURE, see oracle/__init__.py).

theseus/geometry/se2.py (SE2 is implemented in theseus itself, not in torchlie): tensor (...,4) = [x, y, cos, sin],
tangent [u_x, u_y, theta], right perturbations; SO2 pieces from theseus/geometry/so2.py:206-234,278-318.
Thresholds: theseus/global_params.py:46-59 (dtype keyed).  Plain torch ops: autograd differentiates the closed forms,
exactly like the reference (no custom backward anywhere in se2.py / so2.py).
"""
import torch

# theseus/global_params.py:49-50,55-56
EPS = {
    torch.float32: dict(near_zero=3e-2, d_near_zero=1e-1),
    torch.float64: dict(near_zero=1e-6, d_near_zero=1e-3),
}
DOF, NS = 3, 4


def se2_exp(xi):
    """se2.py:239-300 (value part)."""
    eps = EPS[xi.dtype]
    ux, uy, theta = xi[..., 0], xi[..., 1], xi[..., 2]
    cosine, sine = theta.cos(), theta.sin()
    small = theta.abs() < eps["near_zero"]
    one = torch.ones_like(theta)
    theta2, theta3 = theta**2, theta**3
    theta_nz = torch.where(small, one, theta)
    sine_by_theta = torch.where(small, 1 - theta2 / 6, sine / theta_nz)
    cosm1_by_theta = torch.where(small, -theta / 2 + theta3 / 24, (cosine - 1) / theta_nz)
    x = sine_by_theta * ux + cosm1_by_theta * uy
    y = sine_by_theta * uy - cosm1_by_theta * ux
    return torch.stack([x, y, cosine, sine], -1)


def se2_log_jlog(X):
    """se2.py:165-229: returns (xi (...,3), Jlog (...,3,3))."""
    eps = EPS[X.dtype]
    x, y, cosine, sine = X[..., 0], X[..., 1], X[..., 2], X[..., 3]
    theta = torch.atan2(sine, cosine)  # so2.py:220-222
    small = theta.abs() < eps["near_zero"]
    one = torch.ones_like(theta)
    sine_nz = torch.where(small, one, sine)
    h = 0.5 * (1 + cosine) * torch.where(small, 1 + sine**2 / 6, theta / sine_nz)
    half_theta = 0.5 * theta
    ux = h * x + half_theta * y
    uy = h * y - half_theta * x
    theta2 = theta**2
    theta3 = theta * theta2
    dsmall = theta.abs() < eps["d_near_zero"]
    theta_nz = torch.where(dsmall, one, theta)
    omc_nz = torch.where(dsmall, one, 1 - cosine)
    a = torch.where(dsmall, 1 - theta2 / 12.0, half_theta * sine / omc_nz)
    coeff = torch.where(dsmall, theta / 12.0 + theta3 / 720.0, 1.0 / theta_nz - 0.5 * sine / omc_nz)
    z = torch.zeros_like(theta)
    J = torch.stack([
        torch.stack([a, -half_theta, coeff * ux + 0.5 * uy], -1),
        torch.stack([half_theta, a, coeff * uy - 0.5 * ux], -1),
        torch.stack([z, z, one], -1)], -2)
    return torch.stack([ux, uy, theta], -1), J


def se2_adjoint(X):
    """se2.py:309-316: [[c, -s, y], [s, c, -x], [0, 0, 1]]."""
    x, y, c, s = X[..., 0], X[..., 1], X[..., 2], X[..., 3]
    z, o = torch.zeros_like(x), torch.ones_like(x)
    return torch.stack([torch.stack([c, -s, y], -1), torch.stack([s, c, -x], -1), torch.stack([z, z, o], -1)], -2)


def se2_inverse(X):
    """se2.py:334-339: R^-1 = (c, -s); t' = R^-1 (-t)."""
    x, y, c, s = X[..., 0], X[..., 1], X[..., 2], X[..., 3]
    return torch.stack([-(c * x + s * y), -(-s * x + c * y), c, -s], -1)


def se2_compose(A, B):
    """se2.py:318-332, so2.py:224-230."""
    x1, y1, c1, s1 = A[..., 0], A[..., 1], A[..., 2], A[..., 3]
    x2, y2, c2, s2 = B[..., 0], B[..., 1], B[..., 2], B[..., 3]
    return torch.stack([x1 + c1 * x2 - s1 * y2, y1 + s1 * x2 + c1 * y2, c1 * c2 - s1 * s2, s1 * c2 + c1 * s2], -1)


def se2_retract(X, delta):
    """theseus/geometry/lie_group.py:197-198."""
    return se2_compose(X, se2_exp(delta))
