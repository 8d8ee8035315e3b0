// SE3 cost-term VJP shared by the pose-graph (vjp_kernels.hip) and the bundle-adjustment (ba_vjp_kernels.hip) backward
// kernels: gradient of  phi = - sum_r m_r(x, log_radius) s_r^2 (Jlog(E) q)_r log(E)_r ,  E = Z^-1 C, w.r.t. the 12 raw entries of
// Z, the 6 weights s and log_radius (m_r: one factor for the cost, or one per row with flatten_dims) -- see vjp_kernels.hip for the derivation and the torchlie backward semantics it follows.
#pragma once
#include "common.cuh"
#include "dual.cuh"
#include "robust.cuh"

namespace thx {

using D2 = Dual<double>;

template <typename T>
__device__ __forceinline__ void load_se3_any(const T* __restrict__ p, SE3<double>& X) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    X.R[3 * i] = (double)p[4 * i];
    X.R[3 * i + 1] = (double)p[4 * i + 1];
    X.R[3 * i + 2] = (double)p[4 * i + 2];
    X.t[i] = (double)p[4 * i + 3];
  }
}

// grad of phi w.r.t. the 12 entries of Z (row major 3x4) and the 6 weights
__device__ __forceinline__ void cost_vjp(const SE3<double>& Z, const SE3<double>& C, const double* q, const double* s,
                                         const Eps<double>& eps, int loss, double log_radius, double* gZ, double* gs,
                                         double* glr) {
  // value pass
  SE3<double> Zi, E;
  se3_inv(Z, Zi);
  se3_mul(Zi, C, E);
  double xi[6], Jr[9], Jt[9];
  se3_log_jlog(E, eps, xi, Jr, Jt, true);
  double a[6];  // Jlog q  (Jlog = [[Jr, Jt],[0, Jr]])
  {
    double t0[3], t1[3], t2[3];
    mat3_vec(Jr, q, t0);
    mat3_vec(Jt, q + 3, t1);
    mat3_vec(Jr, q + 3, t2);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      a[i] = t0[i] + t1[i];
      a[3 + i] = t2[i];
    }
  }
  // robust factors m_r(x, log_radius) and the plain per-row phi_r (robust.cuh: RobustTerms)
  double phi_r[6], x_r[6], Phi[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    phi_r[r] = -s[r] * s[r] * a[r] * xi[r];
    x_r[r] = (s[r] * xi[r]) * (s[r] * xi[r]);
  }
  RobustTerms<6> rt;
  rt.eval(loss, x_r, log_radius);
  rt.group(phi_r, Phi);
  double gl = 0.0;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    gl += phi_r[r] * rt.m_l[r];
    gs[r] = rt.m[r] * (-2.0 * s[r] * a[r] * xi[r]) + Phi[r] * rt.m_x[r] * (2.0 * s[r] * xi[r] * xi[r]);
  }
  *glr = gl;
  const Eps<D2> epsd{D2(eps.nz), D2(eps.dnz), D2(eps.npi)};
  SE3<D2> Cd;
#pragma unroll
  for (int i = 0; i < 9; ++i) Cd.R[i] = D2(C.R[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) Cd.t[i] = D2(C.t[i]);
  for (int k = 0; k < 12; ++k) {  // run-time loop: one dual evaluation per raw entry of Z
    SE3<D2> Zd, Zid, Ed;
    const int kr = k >> 2, kc = k & 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) Zd.R[3 * i + j] = D2(Z.R[3 * i + j], (i == kr && j == kc) ? 1.0 : 0.0);
      Zd.t[i] = D2(Z.t[i], (i == kr && kc == 3) ? 1.0 : 0.0);
    }
    se3_inv(Zd, Zid);
    se3_mul(Zid, Cd, Ed);
    D2 xid[6], Jrd[9], Jtd[9];
    se3_log_jlog(Ed, epsd, xid, Jrd, Jtd, true);
    // torchlie's log backward: d xi = Jlog [E_R^T dE_t ; vee(E_R^T dE_R) / 2]
    double dR[9], dt[3], M[9], u[6], dxi[6];
#pragma unroll
    for (int i = 0; i < 9; ++i) dR[i] = Ed.R[i].d;
#pragma unroll
    for (int i = 0; i < 3; ++i) dt[i] = Ed.t[i].d;
    mat3_tmul(E.R, dR, M);
    mat3_tvec(E.R, dt, u);
    u[3] = 0.5 * (M[7] - M[5]);
    u[4] = 0.5 * (M[2] - M[6]);
    u[5] = 0.5 * (M[3] - M[1]);
    {
      double t0[3], t1[3], t2[3];
      mat3_vec(Jr, u, t0);
      mat3_vec(Jt, u + 3, t1);
      mat3_vec(Jr, u + 3, t2);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        dxi[i] = t0[i] + t1[i];
        dxi[3 + i] = t2[i];
      }
    }
    // d(Jlog q): dual parts of the Jlog closed forms
    double da[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double top = 0.0, bot = 0.0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        top += Jrd[3 * i + j].d * q[j] + Jtd[3 * i + j].d * q[3 + j];
        bot += Jrd[3 * i + j].d * q[3 + j];
      }
      da[i] = top;
      da[3 + i] = bot;
    }
    double g = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
      g += rt.m[r] * (-s[r] * s[r] * (da[r] * xi[r] + a[r] * dxi[r])) + Phi[r] * rt.m_x[r] * (2.0 * s[r] * s[r] * xi[r] * dxi[r]);
    gZ[k] = g;
  }
}

}  // namespace thx
