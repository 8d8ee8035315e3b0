"""Helpers for the -m gpu parity tests: golden fixture / oracle problem -> device layout."""
import torch

from theseus_amd.compiler import PoseGraphStructure
from theseus_amd.kernels import PGTensors, round_up


def loss_codes(spec, device="cuda"):
    """the oracle's loss spec (oracle/pose_graph.py: PGProblem.robust_*) -> (role code, per-cost int32 table | None)."""
    one = lambda s: 0 if s is None else {"welsch": 1, "huber": 2, "hinge": 3, "gm": 8}[s.split("+")[0]] | (4 if s.endswith("+flatten") else 0)  # noqa: E731
    if spec is None or isinstance(spec, str):
        return one(spec), None
    codes = [one(s) for s in spec]
    return next(c for c in codes if c), torch.tensor(codes, dtype=torch.int32, device=device)


def to_device_problem(p, poses0, device="cuda"):
    """oracle PGProblem (batch-major) -> (PoseGraphStructure, PGTensors) in entity-major device layout."""
    s = PoseGraphStructure.build(p.num_poses, p.edges.tolist(), p.prior_idx.tolist(), dof=p.dof)
    em = lambda t: t.transpose(0, 1).contiguous().to(device)  # noqa: E731  (B,X,...) -> (X,B,...)
    E, Kp = p.edges.shape[0], p.prior_idx.shape[0]
    lr = lambda x, n: None if x is None else em(x.expand(-1, n, 1).to(poses0.dtype))  # noqa: E731
    (rb, tb), (rp, tp) = loss_codes(p.robust_between, device), loss_codes(p.robust_prior, device)
    t = PGTensors(poses=em(poses0), meas=em(p.meas), w_between=em(p.w_between),
                  prior_target=em(p.prior_target), w_prior=em(p.w_prior),
                  robust_between=rb, log_radius_between=lr(p.log_radius_between, E), loss_between=tb,
                  robust_prior=rp, log_radius_prior=lr(p.log_radius_prior, Kp), loss_prior=tp)
    return s, t


def alloc_dense(B, n, dtype, device="cuda"):
    ld = round_up(n, 32)
    H = torch.zeros(B, ld, ld, dtype=dtype, device=device)
    g = torch.zeros(B, n, dtype=dtype, device=device)
    return H, g, ld


def sym_from_lower(H, n):
    Hl = torch.tril(H[:, :n, :n])
    return Hl + torch.tril(Hl, -1).transpose(1, 2)


def factor_and_solve(K, H, n, rhs, damping=None, ellipsoidal=False, eps=1e-8, fused=False):
    """fused=False: thx_chol_factor + thx_chol_solve (cached-factor solve);
    fused=True: thx_chol_factor_forward + thx_chol_solve_backward (what LinearSolver.solve runs)."""
    B, ld = H.shape[0], H.shape[-1]
    nt = (n + 127) // 128
    L = torch.zeros_like(H)
    panels = torch.empty(B, nt, 128, 128, dtype=H.dtype, device=H.device)
    info = torch.empty(B, dtype=torch.int32, device=H.device)
    x = torch.empty_like(rhs)
    if fused:
        y = torch.empty_like(rhs)
        K.chol_factor(H, n, damping, ellipsoidal, eps, L, panels, info, rhs=rhs, y=y)
        K.chol_solve_backward(L, n, panels, y, x)
    else:
        K.chol_factor(H, n, damping, ellipsoidal, eps, L, panels, info)
        K.chol_solve(L, n, panels, rhs, x)
    return L, x, info
