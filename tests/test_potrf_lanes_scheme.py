"""The arithmetic of potrf_inv32_lanes (theseus_amd/csrc/chol_kernels.hip) restated in numpy, operation for operation: the in-register
Cholesky of a 32 x 32 diagonal sub-block works by COLUMN operations on rows held one per lane; the wave's other 32 lanes hold the rows of
the identity and take the same operations, which turns them into L^-T -- column r of W = L^-1 in lane 32 + r, exact zeros above the
diagonal.  No GPU: this pins the scheme (and its fp32 error level) the kernel relies on; the kernel itself is checked against LAPACK
in tests/test_gpu_kernels.py."""
import numpy as np
import pytest


def lanes_scheme(S, dtype):
    """rows 0..31: S, rows 32..63: identity; per pivot c: scale column c by 1/sqrt(d), subtract column c times L[q][c] from column q > c."""
    n = S.shape[0]
    a = np.concatenate([S.astype(dtype), np.eye(n, dtype=dtype)], 0)
    bad = 0
    for c in range(n):
        d = a[c, c]
        if not d > 0:
            bad = bad or c + 1
            d = dtype(1)
        isq = dtype(1) / np.sqrt(d, dtype=dtype)
        a[:, c] = a[:, c] * isq
        for q in range(c + 1, n):
            a[:, q] = a[:, q] - a[:, c] * a[q, c]   # the scalar is lane q's element of column c (v_readlane)
    L = np.tril(a[:n])
    W = a[n:].T   # lane 32 + r holds column r of W
    return L, W, bad


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-4), (np.float64, 1e-12)])
def test_lanes_scheme_gives_factor_and_inverse(dtype, tol):
    rng = np.random.default_rng(3)
    for _ in range(4):
        A = rng.standard_normal((32, 48))
        S = A @ A.T / 48 + 0.05 * np.eye(32)
        L, W, bad = lanes_scheme(S, dtype)
        assert bad == 0
        Lref = np.linalg.cholesky(S)
        scale = np.abs(Lref).max()
        assert np.abs(L - Lref).max() <= tol * scale
        assert np.array_equal(np.triu(W, 1), np.zeros_like(W))   # exact zeros above the diagonal: W is used as a full 32 x 32 operand
        assert np.abs(W.astype(np.float64) @ L.astype(np.float64) - np.eye(32)).max() <= tol * np.linalg.cond(Lref)


def test_lanes_scheme_reports_first_non_positive_pivot():
    S = np.eye(32)
    S[5, 5] = -1.0
    _, _, bad = lanes_scheme(S, np.float64)
    assert bad == 6
