"""The device maths of BackwardMode.UNROLL / TRUNCATED for SE3 pose graphs (theseus_amd/csrc/unroll_se3.cuh) WITHOUT a GPU: the
header is plain C++ over lie.cuh / dual.cuh, compiled here for the host (tests/hostmath: a 6-line shim for <hip/hip_runtime.h>)
and compared with torch autograd through the oracle's formulas -- which carry the reference's autograd conventions (log's
passthrough backward, plain graphs for inverse / compose / adjoint / Jlog) and are pinned to the reference's own unrolled
gradients by tests/test_oracle_golden.py::test_unrolled_gradients_of_a_pose_graph_match_reference."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import lie
from oracle import pose_graph as opg
from tests.conftest import ROOT


@pytest.fixture(scope="module")
def hostmath(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hostmath") / "libhostmath.so")
    src = os.path.join(ROOT, "tests", "hostmath")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", f"-I{src}", f"-I{ROOT}/theseus_amd/csrc", "-Wno-unknown-pragmas",
                    os.path.join(src, "hostmath.cpp"), "-o", out], check=True)
    return ctypes.CDLL(out)


def _ptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _rand_pose(gen, scale_t, scale_r):
    xi = torch.cat([scale_t * (2 * torch.rand(1, 3, dtype=torch.float64, generator=gen) - 1),
                    scale_r * (2 * torch.rand(1, 3, dtype=torch.float64, generator=gen) - 1)], 1)
    return lie.se3_exp(xi)[0]


EPS64 = np.array([lie.EPS[torch.float64][k] for k in ("near_zero", "d_near_zero", "near_pi")])


LOSSES = [(0, None), (1, "welsch"), (2, "huber"), (5, "welsch+flatten"), (6, "huber+flatten"), (3, "hinge"), (7, "hinge+flatten"), (8, "gm"), (12, "gm+flatten")]   # (THX_LOSS_* code, oracle spec)


def _robust(Js, e, spec, log_radius):
    """RobustCostFunction's rescale of one cost (robust_cost_function.py:115-135) through the oracle: (6, n) blocks, (6,) error."""
    if spec is None:
        return Js, e
    Jr, er = opg.robust_rescale([J.view(1, 1, *J.shape) for J in Js], e.view(1, 1, 6), spec if "+" not in spec else [spec], log_radius)
    return [J.view(6, -1) for J in Jr], er.view(6)


@pytest.mark.parametrize("code,spec", LOSSES)
@pytest.mark.parametrize("lam", [0.0, 0.37])     # (> 0: ellipsoidal damping's term -lambda sum_i w_i delta_i H_ii)
@pytest.mark.parametrize("seed,scale_r", [(0, 1.0), (1, 2.5), (2, 0.3), (3, 1e-3)])   # (1e-3: the near-zero Taylor branches)
def test_between_cost_unrolled_vjp_matches_autograd_through_the_oracle(hostmath, seed, scale_r, lam, code, spec):
    gen = torch.Generator().manual_seed(seed)
    Xi, Xj = _rand_pose(gen, 2.0, 1.5), _rand_pose(gen, 2.0, 1.5)
    D = lie.se3_compose(lie.se3_inverse(Xi), Xj)
    Z = lie.se3_compose(D, _rand_pose(gen, 0.2 * scale_r, scale_r))   # E = Z^-1 D has a rotation of ~scale_r
    s = 0.5 + torch.rand(6, dtype=torch.float64, generator=gen)
    wi, wj, di, dj = (torch.randn(6, dtype=torch.float64, generator=gen) for _ in range(4))
    # (log_loss_radius near the squared error, so that Huber's knee and Welsch's decay are both exercised)
    with torch.no_grad():
        x0 = float((opg.between_jac_err(Xi, Xj, Z, s)[2] ** 2).sum())
    lr = torch.tensor([[np.log(max(x0, 1e-12)) - 0.3 + 0.2 * seed]], dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (Xi, Xj, Z, s, lr)]
    J0, J1, e = opg.between_jac_err(leaves[0], leaves[1], leaves[2], leaves[3])
    (J0, J1), e = _robust([J0, J1], e, spec, leaves[4])
    phi = -((J0 @ wi + J1 @ wj) * (e + J0 @ di + J1 @ dj)).sum()
    phi = phi - lam * (((J0 ** 2).sum(0) * wi * di).sum() + ((J1 ** 2).sum(0) * wj * dj).sum())   # -lambda sum_i w_i delta_i (J^T J)_ii
    phi.backward()
    out = np.zeros(43)
    hostmath.hm_edge_vjp.argtypes = [ctypes.POINTER(ctypes.c_double)] * 9 + [ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                                                              ctypes.POINTER(ctypes.c_double)]
    hostmath.hm_edge_vjp(*(_ptr(np.ascontiguousarray(t.detach().numpy().reshape(-1))) for t in (Xi, Xj, Z, s, wi, wj, di, dj)),
                         _ptr(EPS64), lam, code, float(lr), _ptr(out))
    for k, (name, sl) in enumerate((("Xi", slice(0, 12)), ("Xj", slice(12, 24)), ("Z", slice(24, 36)), ("s", slice(36, 42)),
                                    ("log_radius", slice(42, 43)))):
        if leaves[k].grad is None:      # (a plain cost does not depend on log_loss_radius)
            assert name == "log_radius" and out[42] == 0.0
            continue
        want = leaves[k].grad.numpy().reshape(-1)
        np.testing.assert_allclose(out[sl], want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()), err_msg=name)


@pytest.mark.parametrize("code,spec", LOSSES)
@pytest.mark.parametrize("lam", [0.0, 0.37])
@pytest.mark.parametrize("seed,scale_r", [(0, 1.0), (1, 1e-3)])
def test_prior_cost_unrolled_vjp_matches_autograd_through_the_oracle(hostmath, seed, scale_r, lam, code, spec):
    gen = torch.Generator().manual_seed(10 + seed)
    X = _rand_pose(gen, 2.0, 1.5)
    T = lie.se3_compose(X, _rand_pose(gen, 0.2 * scale_r, scale_r))
    s = 0.5 + torch.rand(6, dtype=torch.float64, generator=gen)
    w, d = (torch.randn(6, dtype=torch.float64, generator=gen) for _ in range(2))
    with torch.no_grad():
        x0 = float((opg.local_jac_err(T, X, s)[1] ** 2).sum())
    lr = torch.tensor([[np.log(max(x0, 1e-12)) - 0.2]], dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (X, T, s, lr)]
    J, e = opg.local_jac_err(leaves[1], leaves[0], leaves[2])
    (J,), e = _robust([J], e, spec, leaves[3])
    phi = -((J @ w) * (e + J @ d)).sum() - lam * ((J ** 2).sum(0) * w * d).sum()
    phi.backward()
    out = np.zeros(31)
    hostmath.hm_prior_vjp.argtypes = [ctypes.POINTER(ctypes.c_double)] * 6 + [ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                                                               ctypes.POINTER(ctypes.c_double)]
    hostmath.hm_prior_vjp(*(_ptr(np.ascontiguousarray(t.detach().numpy().reshape(-1))) for t in (X, T, s, w, d)), _ptr(EPS64), lam,
                          code, float(lr), _ptr(out))
    for k, (name, sl) in enumerate((("X", slice(0, 12)), ("T", slice(12, 24)), ("s", slice(24, 30)), ("log_radius", slice(30, 31)))):
        if leaves[k].grad is None:
            assert name == "log_radius" and out[30] == 0.0
            continue
        want = leaves[k].grad.numpy().reshape(-1)
        np.testing.assert_allclose(out[sl], want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()), err_msg=name)


# ---- SE2 / SO3 (theseus_amd/csrc/unroll_g3.cuh) ----------------------------------------------------------------------------------
def _g3_setup(group, gen, scale_r):
    from oracle import lie_se2, lie_so3
    f64 = torch.float64
    rnd3 = lambda a, b_: torch.cat([a * (2 * torch.rand(1, 2, dtype=f64, generator=gen) - 1),      # noqa: E731
                                    b_ * (2 * torch.rand(1, 1, dtype=f64, generator=gen) - 1)], 1)
    if group == "SE2":
        pose = lambda a, b_: lie_se2.se2_exp(rnd3(a, b_))[0]      # noqa: E731
        eps = np.array([lie_se2.EPS[f64]["near_zero"], lie_se2.EPS[f64]["d_near_zero"], 0.0])
        return pose, opg.GROUPS["SE2"], eps, 4, 2
    pose = lambda a, b_: lie_so3.so3_exp(b_ * (2 * torch.rand(1, 3, dtype=f64, generator=gen) - 1))[0]   # noqa: E731
    return pose, opg.GROUPS["SO3"], EPS64, 9, 3


@pytest.mark.parametrize("code,spec", LOSSES)
@pytest.mark.parametrize("lam", [0.0, 0.37])
@pytest.mark.parametrize("seed,scale_r", [(0, 1.0), (1, 2.0), (2, 1e-4)])   # (1e-4: the near-zero Taylor branches of both groups)
@pytest.mark.parametrize("group", ["SE2", "SO3"])
def test_three_dof_groups_unrolled_vjp_matches_autograd_through_the_oracle(hostmath, group, seed, scale_r, lam, code, spec):
    gen = torch.Generator().manual_seed(100 + seed)
    pose, G, eps, NR, gid = _g3_setup(group, gen, scale_r)
    Xi, Xj = pose(2.0, 1.5), pose(2.0, 1.5)
    D = G.compose(G.inverse(Xi), Xj)
    Z = G.compose(D, pose(0.2 * scale_r, scale_r))
    s = 0.5 + torch.rand(3, dtype=torch.float64, generator=gen)
    wi, wj, di, dj = (torch.randn(3, dtype=torch.float64, generator=gen) for _ in range(4))

    def robust(Js, e, lr):
        if spec is None:
            return Js, e
        Jr, er = opg.robust_rescale([J.view(1, 1, *J.shape) for J in Js], e.view(1, 1, 3), spec if "+" not in spec else [spec], lr)
        return [J.view(3, -1) for J in Jr], er.view(3)
    hostmath.hm_g3_vjp.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.POINTER(ctypes.c_double)] * 9 + [
        ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    flat = lambda t: _ptr(np.ascontiguousarray(t.detach().numpy().reshape(-1)))   # noqa: E731
    # ---- Between ----
    with torch.no_grad():
        x0 = float((opg.between_jac_err(Xi, Xj, Z, s, G)[2] ** 2).sum())
    lr = torch.tensor([[np.log(max(x0, 1e-12)) - 0.3 + 0.2 * seed]], dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (Xi, Xj, Z, s, lr)]
    J0, J1, e = opg.between_jac_err(leaves[0], leaves[1], leaves[2], leaves[3], G)
    (J0, J1), e = robust([J0, J1], e, leaves[4])
    phi = -((J0 @ wi + J1 @ wj) * (e + J0 @ di + J1 @ dj)).sum()
    phi = phi - lam * (((J0 ** 2).sum(0) * wi * di).sum() + ((J1 ** 2).sum(0) * wj * dj).sum())
    phi.backward()
    out = np.zeros(3 * NR + 4)
    hostmath.hm_g3_vjp(gid, 1, flat(Xi), flat(Xj), flat(Z), flat(s), flat(wi), flat(wj), flat(di), flat(dj), _ptr(eps), lam, code,
                       float(lr), _ptr(out))
    for k, (name, sl) in enumerate((("Xi", slice(0, NR)), ("Xj", slice(NR, 2 * NR)), ("Z", slice(2 * NR, 3 * NR)),
                                    ("s", slice(3 * NR, 3 * NR + 3)), ("log_radius", slice(3 * NR + 3, 3 * NR + 4)))):
        if leaves[k].grad is None:
            assert name == "log_radius" and out[3 * NR + 3] == 0.0
            continue
        want = leaves[k].grad.numpy().reshape(-1)
        np.testing.assert_allclose(out[sl], want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()), err_msg=f"between {name}")
    # ---- Difference / Local prior ----
    X, T = Xj, G.compose(Xj, pose(0.2 * scale_r, scale_r))
    with torch.no_grad():
        x0 = float((opg.local_jac_err(T, X, s, G)[1] ** 2).sum())
    lr = torch.tensor([[np.log(max(x0, 1e-12)) - 0.2]], dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (X, T, s, lr)]
    J, e = opg.local_jac_err(leaves[1], leaves[0], leaves[2], G)
    (J,), e = robust([J], e, leaves[3])
    phi = -((J @ wj) * (e + J @ dj)).sum() - lam * ((J ** 2).sum(0) * wj * dj).sum()
    phi.backward()
    out = np.zeros(3 * NR + 4)
    hostmath.hm_g3_vjp(gid, 0, flat(X), flat(X), flat(T), flat(s), flat(wi), flat(wj), flat(di), flat(dj), _ptr(eps), lam, code,
                       float(lr), _ptr(out))
    for k, (name, sl) in enumerate((("X", slice(NR, 2 * NR)), ("T", slice(2 * NR, 3 * NR)), ("s", slice(3 * NR, 3 * NR + 3)),
                                    ("log_radius", slice(3 * NR + 3, 3 * NR + 4)))):
        if leaves[k].grad is None:
            continue
        want = leaves[k].grad.numpy().reshape(-1)
        np.testing.assert_allclose(out[sl], want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()), err_msg=f"prior {name}")


# ---- bundle adjustment (theseus_amd/csrc/unroll_ba.cuh) -----------------------------------------------------------------------
def _robust2(Js, e, spec, log_radius):
    if spec is None:
        return Js, e
    Jr, er = opg.robust_rescale([J.view(1, 1, *J.shape) for J in Js], e.view(1, 1, 2), spec if "+" not in spec else [spec], log_radius)
    return [J.view(2, -1) for J in Jr], er.view(2)


@pytest.mark.parametrize("code,spec", LOSSES)
@pytest.mark.parametrize("lam", [0.0, 0.37])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_reprojection_cost_unrolled_vjp_matches_autograd_through_the_oracle(hostmath, seed, lam, code, spec):
    """Reprojection (reprojection.py:54-94) in thx_ba_unroll_vjp: camera, point, feature, weights, calibration, log_loss_radius."""
    from oracle import ba as oba
    f64 = torch.float64
    gen = torch.Generator().manual_seed(200 + seed)
    cam = _rand_pose(gen, 0.5, 0.4)
    X = torch.randn(3, dtype=f64, generator=gen) * 0.5
    X = lie.se3_inverse(cam)[:, :3] @ (torch.tensor([0.3, -0.2, -4.0], dtype=f64) + X * 0.3 - cam[:, 3])   # in front of the camera
    calib = torch.tensor([500.0 + 30 * seed, -0.05 + 0.02 * seed, 0.01 * seed], dtype=f64)
    s = 0.5 + torch.rand(2, dtype=f64, generator=gen)
    wc, dc = (torch.randn(6, dtype=f64, generator=gen) * 0.05 for _ in range(2))
    wp, dp = (torch.randn(3, dtype=f64, generator=gen) * 0.05 for _ in range(2))
    with torch.no_grad():
        e0 = oba.reprojection_jac_err(cam, X, torch.zeros(2, dtype=f64), calib[0:1], calib[1:2], calib[2:3])[2]
    feat = e0 + torch.randn(2, dtype=f64, generator=gen) * 3.0          # a residual of a few pixels
    x0 = float(((e0 - feat) * s).pow(2).sum())
    lr = torch.tensor([[np.log(max(x0, 1e-12)) - 0.3 + 0.2 * seed]], dtype=f64)
    leaves = [t.clone().requires_grad_(True) for t in (cam, X, feat, s, calib, lr)]
    c_ = leaves[4]
    Jc, Jp, e = oba.reprojection_jac_err(leaves[0], leaves[1], leaves[2], c_[0:1], c_[1:2], c_[2:3])
    Jc, Jp, e = Jc * leaves[3].unsqueeze(-1), Jp * leaves[3].unsqueeze(-1), e * leaves[3]
    (Jc, Jp), e = _robust2([Jc, Jp], e, spec, leaves[5])
    phi = -((Jc @ wc + Jp @ wp) * (e + Jc @ dc + Jp @ dp)).sum()
    phi = phi - lam * (((Jc ** 2).sum(0) * wc * dc).sum() + ((Jp ** 2).sum(0) * wp * dp).sum())
    phi.backward()
    out = np.zeros(23)
    hostmath.hm_reproj_vjp.argtypes = [ctypes.POINTER(ctypes.c_double)] * 9 + [ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                                                                ctypes.POINTER(ctypes.c_double)]
    hostmath.hm_reproj_vjp(*(_ptr(np.ascontiguousarray(t.detach().numpy().reshape(-1))) for t in (cam, X, feat, calib, s, wc, wp, dc, dp)),
                           lam, code, float(lr), _ptr(out))
    slices = (("cam", 0, slice(0, 12)), ("X", 1, slice(12, 15)), ("feat", 2, slice(15, 17)), ("s", 3, slice(17, 19)),
              ("calib", 4, slice(19, 22)), ("log_radius", 5, slice(22, 23)))
    for name, k, sl in slices:
        if leaves[k].grad is None:
            assert name == "log_radius" and out[22] == 0.0
            continue
        want = leaves[k].grad.numpy().reshape(-1)
        np.testing.assert_allclose(out[sl], want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()), err_msg=name)


@pytest.mark.parametrize("lam", [0.0, 0.37])
def test_point_prior_unrolled_vjp_matches_autograd(hostmath, lam):
    """Point3 Difference (vector.py:150-178: e = x - target, J = I) in thx_ba_unroll_vjp."""
    f64 = torch.float64
    gen = torch.Generator().manual_seed(5)
    X, T, s, w, d = (torch.randn(3, dtype=f64, generator=gen) for _ in range(5))
    leaves = [t.clone().requires_grad_(True) for t in (X, T, s)]
    J = torch.diag_embed(leaves[2])
    e = (leaves[0] - leaves[1]) * leaves[2]
    phi = -((J @ w) * (e + J @ d)).sum() - lam * ((J ** 2).sum(0) * w * d).sum()
    phi.backward()
    out = np.zeros(9)
    hostmath.hm_pt_prior_vjp.argtypes = [ctypes.POINTER(ctypes.c_double)] * 5 + [ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    hostmath.hm_pt_prior_vjp(*(_ptr(np.ascontiguousarray(t.detach().numpy())) for t in (X, T, s, w, d)), lam, _ptr(out))
    for k, sl in enumerate((slice(0, 3), slice(3, 6), slice(6, 9))):
        np.testing.assert_allclose(out[sl], leaves[k].grad.numpy(), rtol=0, atol=1e-12)
