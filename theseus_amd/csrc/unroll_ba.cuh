// BackwardMode.UNROLL / TRUNCATED on bundle-adjustment objectives (theseus/optimizer/nonlinear/nonlinear_least_squares.py:223-292;
// the iteration delta = (H + D)^-1 g with H = A^T A and g = A^T b both in the graph -- unroll_se3.cuh has the derivation):
// with w = (H + D)^-1 dL/d delta (one solve with that iteration's Schur factor) every cost contributes the gradient of
//     phi = - sum_r s_r^2 [ (J w)_r (eps_r + (J delta)_r)  +  lambda sum_k J_rk^2 w_k delta_k ]        (w, delta held constant;
//                                                                                         lambda: ellipsoidal damping only)
// Reprojection cost (theseus/embodied/measurements/reprojection.py:54-94): raw residual eps (2) and Jacobian J (2 x 9 =
// [camera 6 | point 3]) are functions of the camera's 3 x 4 entries, the point, the feature and the calibration; all of the
// reference's graph is plain autograd here (SE3.transform_from's backward is the matrix rule, torchlie/functional/se3_impl.py:
// 779-800; its Jacobians [R | -R hat(X)], R are plain tensor expressions, se3_impl.py:764-777), so phi is differentiated in
// forward mode (Dual<double>) along one raw input entry at a time, with no convention to reproduce.
// RobustCostFunction (not detached, robust_cost_function.py:115-135): Phi = sum_r m_r phi_r, m = rho'(x) + eps,
// x_r = (s_r eps_r)^2.  Camera priors are SE3 Difference costs (unroll_prior_vjp of unroll_se3.cuh); Point3 Difference priors
// (vector.py:150-178: e = x - target, J = I) are closed forms.
// Plain C++ templates: also compiled for the HOST by tests/hostmath.
#pragma once
#include "unroll_se3.cuh"

namespace thx {

// phi_r, x_r (2 rows) of one Reprojection cost on scalar type S; a_out / b_out / ell_out / eps_out (may be null) receive
// (J w)_r, eps_r + (J delta)_r, sum_k J_rk^2 w_k delta_k and eps_r as plain doubles.
__device__ __forceinline__ double unroll_val(double x) { return x; }
__device__ __forceinline__ double unroll_val(const UD& x) { return x.v; }

template <typename S>
__device__ __forceinline__ void unroll_reproj_phi(const SE3<S>& cam, const S* X, const S* feat, S f, S k1, S k2, const double* s,
                                                  const double* wc, const double* wp, const double* dc, const double* dp,
                                                  double lam, S* phi_r, S* x_r, double* a_out, double* b_out, double* ell_out,
                                                  double* eps_out) {
  S pc[3];
  mat3_vec(cam.R, X, pc);
#pragma unroll
  for (int i = 0; i < 3; ++i) pc[i] = pc[i] + cam.t[i];
  const S iz = S(1.0) / pc[2];
  const S proj[2] = {S(0.0) - pc[0] * iz, S(0.0) - pc[1] * iz};
  const S q = proj[0] * proj[0] + proj[1] * proj[1];
  const S factor = f * (S(1.0) + q * (k1 + q * k2));
  const S dfactor = f * (k1 + S(2.0) * q * k2);
  const S eps[2] = {proj[0] * factor - feat[0], proj[1] * factor - feat[1]};
  // d pc / d [camera tangent (lin, ang) | point] = [R | -R hat(X) | R]: column k as a 3-vector
  S a[2] = {S(0.0), S(0.0)}, c[2] = {S(0.0), S(0.0)}, ell[2] = {S(0.0), S(0.0)};
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    S col[3];
    if (k < 3 || k >= 6) {
      const int j = k < 3 ? k : k - 6;
#pragma unroll
      for (int i = 0; i < 3; ++i) col[i] = cam.R[3 * i + j];
    } else {
      // -R hat(X) e_j = -R (X x e_j)
      const int j = k - 3;
      S e[3] = {S(j == 0 ? 1.0 : 0.0), S(j == 1 ? 1.0 : 0.0), S(j == 2 ? 1.0 : 0.0)}, xe[3], rx[3];
      cross3(X, e, xe);
      mat3_vec(cam.R, xe, rx);
#pragma unroll
      for (int i = 0; i < 3; ++i) col[i] = S(0.0) - rx[i];
    }
    const S j2 = col[2] * iz;
    const S pj0 = (pc[0] * j2 - col[0]) * iz, pj1 = (pc[1] * j2 - col[1]) * iz;
    const S qj = S(2.0) * (proj[0] * pj0 + proj[1] * pj1);
    const S J0 = pj0 * factor + proj[0] * qj * dfactor, J1 = pj1 * factor + proj[1] * qj * dfactor;
    const double wk = k < 6 ? wc[k] : wp[k - 6], dk = k < 6 ? dc[k] : dp[k - 6];
    a[0] = a[0] + J0 * S(wk);
    a[1] = a[1] + J1 * S(wk);
    c[0] = c[0] + J0 * S(dk);
    c[1] = c[1] + J1 * S(dk);
    ell[0] = ell[0] + J0 * J0 * S(wk * dk);
    ell[1] = ell[1] + J1 * J1 * S(wk * dk);
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const S bsum = eps[r] + c[r];
    phi_r[r] = S(0.0) - S(s[r] * s[r]) * (a[r] * bsum + S(lam) * ell[r]);
    x_r[r] = S(s[r] * s[r]) * (eps[r] * eps[r]);
    if (a_out) a_out[r] = unroll_val(a[r]);
    if (b_out) b_out[r] = unroll_val(bsum);
    if (ell_out) ell_out[r] = unroll_val(ell[r]);
    if (eps_out) eps_out[r] = unroll_val(eps[r]);
  }
}

// Gradients of one Reprojection cost's Phi: gcam (12 raw entries), gX (3), gfeat (2), gs (2 weights), gcal (focal, k1, k2), glr
__device__ __forceinline__ void unroll_reproj_vjp(const SE3<double>& cam, const double* X, const double* feat, double f, double k1,
                                                  double k2, const double* s, const double* wc, const double* wp, const double* dc,
                                                  const double* dp, double lam, int loss, double log_radius, double* gcam,
                                                  double* gX, double* gfeat, double* gs, double* gcal, double* glr) {
  double phi[2], x[2], a[2], b[2], ell[2], ep[2], P[2];
  unroll_reproj_phi<double>(cam, X, feat, f, k1, k2, s, wc, wp, dc, dp, lam, phi, x, a, b, ell, ep);
  RobustTerms<2> rt;
  rt.eval(loss, x, log_radius);
  rt.group(phi, P);
  *glr = phi[0] * rt.m_l[0] + phi[1] * rt.m_l[1];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    // phi_r = -s_r^2 (a_r b_r + lambda ell_r), b_r = eps_r + c_r, eps_r = pi_r - u_r;  x_r = s_r^2 eps_r^2
    gs[r] = rt.m[r] * (-2.0 * s[r] * (a[r] * b[r] + lam * ell[r])) + P[r] * rt.m_x[r] * (2.0 * s[r] * ep[r] * ep[r]);
    gfeat[r] = rt.m[r] * (s[r] * s[r] * a[r]) - P[r] * rt.m_x[r] * (2.0 * s[r] * s[r] * ep[r]);
  }
  SE3<UD> C;
  UD Xd[3], fd[2] = {UD(feat[0]), UD(feat[1])}, phid[2], xd[2];
  for (int k = 0; k < 18; ++k) {   // run-time loop: one dual evaluation per raw entry of the camera, the point, the calibration
    unroll_seed(cam, k < 12 ? k : -1, C);
#pragma unroll
    for (int i = 0; i < 3; ++i) Xd[i] = UD(X[i], k == 12 + i ? 1.0 : 0.0);
    unroll_reproj_phi<UD>(C, Xd, fd, UD(f, k == 15 ? 1.0 : 0.0), UD(k1, k == 16 ? 1.0 : 0.0), UD(k2, k == 17 ? 1.0 : 0.0), s, wc, wp,
                          dc, dp, lam, phid, xd, nullptr, nullptr, nullptr, nullptr);
    const double g = rt.m[0] * phid[0].d + P[0] * rt.m_x[0] * xd[0].d + rt.m[1] * phid[1].d + P[1] * rt.m_x[1] * xd[1].d;
    if (k < 12) gcam[k] = g;
    else if (k < 15) gX[k - 12] = g;
    else gcal[k - 15] = g;
  }
}

// Point3 Difference prior: e = s (X - t), J = diag(s):  phi = - sum_r s_r^2 w_r (X_r - t_r + delta_r + lambda delta_r)
__device__ __forceinline__ void unroll_pt_prior_vjp(const double* X, const double* t, const double* s, const double* w, const double* d,
                                                    double lam, double* gX, double* gT, double* gs) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    gX[r] = -s[r] * s[r] * w[r];
    gT[r] = s[r] * s[r] * w[r];
    gs[r] = -2.0 * s[r] * w[r] * (X[r] - t[r] + d[r] + lam * d[r]);
  }
}

}  // namespace thx
