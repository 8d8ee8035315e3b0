"""-m gpu: implicit backward mode through TheseusLayer on the HIP kernels (thx_se3_retract_vjp, thx_chol_solve with
the cached factor, thx_pg_vjp) against gradients recorded from the REAL reference."""
import dataclasses

import numpy as np
import pytest
import torch

from oracle import pose_graph as opg
from tests.helpers import golden_problem, load_golden
from tests.implicit_common import check_against_reference, run_implicit

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["pg_f64_implicit", "pg_f64_implicit_b", "pg2_f64_implicit", "pg3_f64_implicit"])
def test_implicit_gradients_match_reference(name):
    import theseus_amd as th
    g = load_golden(name)
    final, loss, grads, info, _, _ = run_implicit(th, g, "cuda")
    check_against_reference(g, final, loss, grads)


def _riemannian(X, G):
    """Tangent projection of a gradient w.r.t. the raw 3x4 entries of X: [R skew(R^T G_R) | G_t].  The component
    normal to SO(3) depends on how a closed form extends off the manifold -- torchlie's Taylor and exact branches
    extend differently, and fp32 / fp64 switch branches at different angles (global_params.py:44-58) -- so only the
    projected gradient is comparable ACROSS dtypes (the reference's own fp32 and fp64 raw gradients differ)."""
    R = X[..., :3]
    M = R.transpose(-1, -2) @ G[..., :3]
    return torch.cat([R @ (0.5 * (M - M.transpose(-1, -2))), G[..., 3:]], -1)


def test_implicit_fp32_gradients_close_to_fp64_reference():
    """fp32 storage, fp64-register VJP: gradients of the fp32 run against the reference's fp64 gradients of the same
    problem.  The forward solve is an fp32 Cholesky of a gauge-weak system, so the tolerance is the forward one
    (5e-3 of the gradient scale), not rounding."""
    import theseus_amd as th
    g = load_golden("pg_f64_implicit")
    g32 = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 else v) for k, v in g.items()}
    final, loss, grads, _, _, _ = run_implicit(th, g32, "cuda")
    assert np.abs(final.numpy() - g["final"]).max() < 2e-3
    for key, ref in (("meas", "grad_meas"), ("w_between", "grad_w_between"), ("prior_target", "grad_prior_target"),
                     ("w_prior", "grad_w_prior")):
        want, got = torch.from_numpy(g[ref]), grads[key].double()
        if key in ("meas", "prior_target"):
            X = torch.from_numpy(g[key])
            want, got = _riemannian(X, want), _riemannian(X, got)
        assert (got - want).abs().max() <= 5e-3 * want.abs().max(), key


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_vjp_kernels_vs_oracle_autograd(dtype):
    """thx_pg_vjp / thx_se3_retract_vjp on their own against torch autograd through the oracle (which is pinned to
    the reference's gradient conventions by tests/test_oracle_golden.py)."""
    from tests.gpu_helpers import to_device_problem
    from theseus_amd.kernels import default_kernels
    K = default_kernels()
    import contextlib
    from tests.helpers import f32_thresholds
    g = load_golden("pg_f64_implicit_b")
    p, poses0, _ = golden_problem(g)
    if dtype == torch.float32:  # same values as the kernel sees; the checker switches Taylor branches where fp32 does
        r32 = lambda x: x.float().double()  # noqa: E731
        p = dataclasses.replace(p, meas=r32(p.meas), w_between=r32(p.w_between), prior_target=r32(p.prior_target),
                                w_prior=r32(p.w_prior))
        poses0 = r32(poses0)
    ctx = f32_thresholds() if dtype == torch.float32 else contextlib.nullcontext()
    B, n = poses0.shape[0], 6 * p.num_poses
    gen = torch.Generator().manual_seed(3)
    w = torch.randn(B, n, dtype=torch.float64, generator=gen)
    full = lambda a: a.expand(B, *a.shape[1:]).clone().requires_grad_(True)  # noqa: E731
    leaves = [full(p.meas), full(p.w_between), full(p.prior_target), full(p.w_prior)]
    pg = dataclasses.replace(p, meas=leaves[0], w_between=leaves[1], prior_target=leaves[2], w_prior=leaves[3])
    if dtype == torch.float32:
        w = w.float().double()
    with ctx:
        A, b = opg.dense_linearize(pg, poses0)
        _, Atb = opg.hessian(A, b)
        ref = torch.autograd.grad((w * Atb.squeeze(2)).sum(), leaves)
    cast = lambda x: x.to(dtype)  # noqa: E731
    p_d = dataclasses.replace(p, meas=cast(p.meas), w_between=cast(p.w_between), prior_target=cast(p.prior_target),
                              w_prior=cast(p.w_prior))
    s, t = to_device_problem(p_d, cast(poses0))
    E, Kp = s.num_edges, s.num_priors
    outs = [torch.empty(E, B, 3, 4, dtype=dtype, device="cuda"), torch.empty(E, B, 6, dtype=dtype, device="cuda"),
            torch.empty(Kp, B, 3, 4, dtype=dtype, device="cuda"), torch.empty(Kp, B, 6, dtype=dtype, device="cuda")]
    K.pg_vjp(s.on("cuda"), t, cast(w).cuda(), *outs)
    tol = 1e-9 if dtype == torch.float64 else 5e-7  # fp32: same inputs, fp64 registers, output rounded once
    for got, want in zip(outs, ref):
        want = want.transpose(0, 1)
        assert (got.cpu().double() - want).abs().max() <= tol * want.abs().max()
    # retract VJP
    delta = 0.3 * torch.randn(B, n, dtype=torch.float64, generator=gen)
    gout = torch.randn(B, p.num_poses, 3, 4, dtype=torch.float64, generator=gen)
    if dtype == torch.float32:
        delta, gout = delta.float().double(), gout.float().double()
    d = delta.clone().requires_grad_(True)
    with ctx:
        (gref,) = torch.autograd.grad(opg.retract(poses0, d * 0.75), d, gout)
    gd = torch.empty(B, n, dtype=dtype, device="cuda")
    K.se3_retract_vjp(cast(poses0).transpose(0, 1).contiguous().cuda(), cast(delta).cuda(), 0.75,
                      cast(gout).transpose(0, 1).contiguous().cuda(), gd)
    assert (gd.cpu().double() - gref).abs().max() <= (1e-10 if dtype == torch.float64 else 5e-7) * gref.abs().max()
