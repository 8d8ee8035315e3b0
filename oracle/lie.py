"""Oracle: SO3/SE3 arithmetic restated from torchlie (TEST INFRASTRUCTURE, see oracle/__init__.py).

Conventions (torchlie/torchlie/functional/se3_impl.py:195-196,387): SE3 tensor (...,3,4) = [R | t];
tangent xi = [v(3), w(3)] (linear first); right perturbations.  Thresholds are dtype keyed
(torchlie/torchlie/global_params.py:44-58).  No re-normalisation of R anywhere.
"""
import torch

# torchlie/torchlie/global_params.py:44-58
EPS = {
    torch.float32: dict(near_zero=1e-2, d_near_zero=2e-1, near_pi=1e-2),
    torch.float64: dict(near_zero=5e-3, d_near_zero=1e-2, near_pi=1e-7),
}


def _hat(w):
    """so3_impl.py:587-599  hat(w)[0,1]=-w2, [0,2]=w1, [1,2]=-w0."""
    z = torch.zeros_like(w[..., 0])
    return torch.stack(
        [
            torch.stack([z, -w[..., 2], w[..., 1]], -1),
            torch.stack([w[..., 2], z, -w[..., 0]], -1),
            torch.stack([-w[..., 1], w[..., 0], z], -1),
        ],
        -2,
    )


def _outer(a, b):
    return a.unsqueeze(-1) * b.unsqueeze(-2)


def so3_exp_helper(w):
    """so3_impl.py:220-261."""
    eps = EPS[w.dtype]
    theta = torch.linalg.norm(w, dim=-1)
    theta2 = theta**2
    nz = theta < eps["near_zero"]
    one = torch.ones_like(theta)
    theta_nz = torch.where(nz, one, theta)
    theta2_nz = torch.where(nz, one, theta2)
    cosine = torch.where(nz, 8 / (4 + theta2) - 1, theta.cos())
    sine = theta.sin()
    A = torch.where(nz, 0.5 * cosine + 0.5, sine / theta_nz)
    Bc = torch.where(nz, 0.5 * A, (1 - cosine) / theta2_nz)
    R = Bc[..., None, None] * _outer(w, w)
    R = R + cosine[..., None, None] * torch.eye(3, dtype=w.dtype)
    R = R + _hat(A[..., None] * w)
    return R, dict(theta=theta, theta2=theta2, theta_nz=theta_nz, theta2_nz=theta2_nz, sine=sine,
                   cosine=cosine, A=A, B=Bc, nz=nz)


def so3_exp(w):
    return so3_exp_helper(w)[0]


def so3_log_helper(R):
    """so3_impl.py:366-433 (sine-axis, atan2, near-zero / near-pi branches)."""
    eps = EPS[R.dtype]
    sa = torch.stack(
        [
            0.5 * (R[..., 2, 1] - R[..., 1, 2]),
            0.5 * (R[..., 0, 2] - R[..., 2, 0]),
            0.5 * (R[..., 1, 0] - R[..., 0, 1]),
        ],
        -1,
    )
    cosine = 0.5 * (R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2] - 1)
    sine = torch.linalg.norm(sa, dim=-1)
    theta = torch.atan2(sine, cosine)
    nz = theta < eps["near_zero"]
    npi = (1 + cosine) <= eps["near_pi"]
    nznp = nz | npi
    sine_nz = torch.where(nznp, torch.ones_like(sine), sine)
    scale = torch.where(nznp, 1 + sine**2 / 6, theta / sine_nz)
    ret = sa * scale[..., None]
    # near pi: major diagonal
    d0, d1, d2 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    major = ((d1 > d0) & (d1 > d2)).long() + 2 * ((d2 > d0) & (d2 > d1)).long()
    idx = major[..., None, None].expand(*major.shape, 1, 3)
    row = torch.gather(R, -2, idx).squeeze(-2)  # R[m, :]
    idxc = major[..., None, None].expand(*major.shape, 3, 1)
    col = torch.gather(R, -1, idxc).squeeze(-1)  # R[:, m]
    sel = 0.5 * (row + col)
    onehot = torch.nn.functional.one_hot(major, 3).to(R.dtype)
    sel = sel - onehot * cosine[..., None]
    nrm = torch.where(nz, torch.ones_like(sine), torch.linalg.norm(sel, dim=-1))
    axis = sel / nrm[..., None]
    sgn_tmp = torch.sign((sa * onehot).sum(-1))
    sgn = torch.where(sgn_tmp != 0, sgn_tmp, torch.ones_like(sgn_tmp))
    w = torch.where(npi[..., None], axis * (theta * sgn)[..., None], ret)
    return w, dict(theta=theta, sine=sine, cosine=cosine)


def so3_log(R):
    return so3_log_helper(R)[0]


def so3_jlog_helper(w, theta, sine, cosine):
    """so3_impl.py:442-479: J = b w w^T + a I + 0.5 hat(w), s,c taken from the matrix."""
    eps = EPS[w.dtype]
    dnz = theta < eps["d_near_zero"]
    theta2 = theta**2
    st = sine * theta
    tcm2 = 2 * cosine - 2
    one = torch.ones_like(theta)
    tcm2_nz = torch.where(dnz, one, tcm2)
    theta2_nz = torch.where(dnz, one, theta2)
    a = torch.where(dnz, 1 - theta2 / 12, -st / tcm2_nz)
    b = torch.where(dnz, 1.0 / 12 + theta2 / 720, (st + tcm2) / (theta2_nz * tcm2_nz))
    bw = b[..., None] * w
    J = _outer(bw, w) + 0.5 * _hat(w) + a[..., None, None] * torch.eye(3, dtype=w.dtype)
    return J, bw


def se3_exp(xi):
    """se3_impl.py:178-216."""
    eps = EPS[xi.dtype]
    v, w = xi[..., :3], xi[..., 3:]
    R, h = so3_exp_helper(w)
    theta, theta2 = h["theta"], h["theta2"]
    nz = theta < eps["near_zero"]
    theta3_nz = h["theta_nz"] * h["theta2_nz"]
    Ct = torch.where(nz, 1.0 / 6 - theta2 / 120, (theta - h["sine"]) / theta3_nz)
    t = h["A"][..., None] * v
    t = t + h["B"][..., None] * torch.linalg.cross(w, v, dim=-1)
    t = t + Ct[..., None] * w * (w * v).sum(-1, keepdim=True)
    return torch.cat([R, t[..., None]], -1)


def se3_log_helper(X):
    """se3_impl.py:354-396 (value path switches on near_zero)."""
    eps = EPS[X.dtype]
    R, t = X[..., :3], X[..., 3]
    w, h = so3_log_helper(R)
    theta, sine, cosine = h["theta"], h["sine"], h["cosine"]
    nz = theta < eps["near_zero"]
    theta2 = theta**2
    st = sine * theta
    tcm2 = 2 * cosine - 2
    one = torch.ones_like(theta)
    tcm2_nz = torch.where(nz, one, tcm2)
    theta2_nz = torch.where(nz, one, theta2)
    a = torch.where(nz, 1 - theta2 / 12, -st / tcm2_nz)
    b = torch.where(nz, 1.0 / 12 + theta2 / 720, (st + tcm2) / (theta2_nz * tcm2_nz))
    v = a[..., None] * t
    v = v - 0.5 * torch.linalg.cross(w, t, dim=-1)
    v = v + b[..., None] * w * (w * t).sum(-1, keepdim=True)
    xi = torch.cat([v, w], -1)
    return xi, dict(theta=theta, theta2=theta2, theta2_nz=theta2_nz, sine=sine, cosine=cosine,
                    tcm2_nz=tcm2_nz)


def se3_log(X):
    return se3_log_helper(X)[0]


def se3_log_jlog(X):
    """se3_impl.py:405-457: returns (xi, Jlog(6x6))."""
    eps = EPS[X.dtype]
    xi, h = se3_log_helper(X)
    v, w = xi[..., :3], xi[..., 3:]
    theta, theta2, sine, cosine = h["theta"], h["theta2"], h["sine"], h["cosine"]
    dnz = theta < eps["d_near_zero"]
    Jr, bw = so3_jlog_helper(w, theta, sine, cosine)
    one = torch.ones_like(theta)
    theta_nz = torch.where(dnz, one, theta)
    theta4_nz = h["theta2_nz"] ** 2
    tcm2_nz = h["tcm2_nz"]
    c = torch.where(dnz, -1 / 360.0 - theta2 / 7560.0,
                    -(2 * tcm2_nz + theta * sine + theta2) / (theta4_nz * tcm2_nz))
    d = torch.where(dnz, -1 / 6.0 - theta2 / 180.0, (theta - sine) / (theta_nz * tcm2_nz))
    e = (w * v).sum(-1)
    Jt = _outer((c * e)[..., None] * w, w) + _outer(bw, v) + _outer(v, bw)
    Jt = Jt + (e * d)[..., None, None] * torch.eye(3, dtype=X.dtype) + 0.5 * _hat(v)
    top = torch.cat([Jr, Jt], -1)
    bot = torch.cat([torch.zeros_like(Jr), Jr], -1)
    return xi, torch.cat([top, bot], -2)


def se3_adjoint(X):
    """se3_impl.py:531-538: [[R, hat(t) R],[0, R]]."""
    R, t = X[..., :3], X[..., 3]
    top = torch.cat([R, _hat(t) @ R], -1)
    bot = torch.cat([torch.zeros_like(R), R], -1)
    return torch.cat([top, bot], -2)


def se3_inverse(X):
    """se3_impl.py:578-581."""
    Rt = X[..., :3].transpose(-1, -2)
    return torch.cat([Rt, -(Rt @ X[..., 3:])], -1)


def se3_compose(X0, X1):
    """se3_impl.py:703-708."""
    R = X0[..., :3] @ X1[..., :3]
    t = X0[..., :3] @ X1[..., 3:] + X0[..., 3:]
    return torch.cat([R, t], -1)


def se3_retract(X, delta):
    """theseus/geometry/lie_group.py:197-198: X . exp(delta)."""
    return se3_compose(X, se3_exp(delta))


class _LogPassthrough(torch.autograd.Function):
    """What autograd sees when the reference asks ``SE3.log(X, jacobians=[...])``
    (torchlie/functional/lie_group.py:60-84,148-155): the VALUE is the formula's, the BACKWARD is
    ``_log_backward`` (se3_impl.py:487-493): grad_X = R @ lift(J^T g * [1,1,1,.5,.5,.5]) -- the tangent
    projection, not the derivative of the closed form; the Jacobian itself keeps its plain autograd graph."""

    @staticmethod
    def forward(ctx, X, xi, J):
        ctx.save_for_backward(X, J)
        return xi.clone()

    @staticmethod
    def backward(ctx, g):
        X, J = ctx.saved_tensors
        h = (J.transpose(-1, -2) @ g.unsqueeze(-1)).squeeze(-1)
        lifted = torch.cat([_hat(0.5 * h[..., 3:]), h[..., :3, None]], -1)  # se3_impl.py:876-882
        return X[..., :3] @ lifted, None, None


def se3_log_jlog_autograd(X):
    """se3_log_jlog with the reference's autograd semantics (see _LogPassthrough); same values."""
    xi, J = se3_log_jlog(X)
    return _LogPassthrough.apply(X, xi, J), J
