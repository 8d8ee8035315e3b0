// BackwardMode.UNROLL / TRUNCATED on SE2 / SO3 pose graphs: the 3-dof twin of unroll_se3.cuh (which states the maths).  One
// generic evaluation of a cost's phi_r / x_r on dual numbers over a small group adapter:
//   UG_SE2 -- theseus/geometry/se2.py: plain autograd through every closed form (lie_se2.cuh on Dual<double>);
//   UG_SO3 -- torchlie's SO3 (so3_impl.py): log has the passthrough backward d log = Jlog vee2(E^T dE) / 2 (:489-496), inverse /
//             compose / the Jlog closed forms are plain; Ad(R) = R.
// Plain C++ templates: also compiled for the host by tests/hostmath.
#pragma once
#include "dual.cuh"
#include "lie_se2.cuh"
#include "lie_so3.cuh"
#include "robust.cuh"

namespace thx {

using UD3 = Dual<double>;

struct UG_SE2 {
  static constexpr int NR = 4;   // raw entries [x, y, cos, sin]
  template <typename S>
  struct X {
    SE2<S> g;
  };
  using EpsD = Eps2<double>;
  using EpsU = Eps2<UD3>;
  static __device__ __forceinline__ EpsU lift(const EpsD& e) { return EpsU{UD3(e.nz), UD3(e.dnz)}; }
  static __device__ __forceinline__ void seed(const double* raw, int k, X<UD3>& y) {
    y.g = SE2<UD3>{UD3(raw[0], k == 0 ? 1.0 : 0.0), UD3(raw[1], k == 1 ? 1.0 : 0.0), UD3(raw[2], k == 2 ? 1.0 : 0.0),
                   UD3(raw[3], k == 3 ? 1.0 : 0.0)};
  }
  static __device__ __forceinline__ void inv(const X<UD3>& a, X<UD3>& y) { se2_inv(a.g, y.g); }
  static __device__ __forceinline__ void mul(const X<UD3>& a, const X<UD3>& b, X<UD3>& z) { se2_mul(a.g, b.g, z.g); }
  static __device__ __forceinline__ void adjoint(const X<UD3>& a, UD3* A) { se2_adjoint(a.g, A); }
  static __device__ __forceinline__ void log_jlog(const X<UD3>& E, const EpsU& eps, UD3* xi, UD3* J) {
    se2_log_jlog(E.g, eps, xi, J, true);   // plain autograd in the reference
  }
};

struct UG_SO3 {
  static constexpr int NR = 9;   // raw entries of R, row major
  template <typename S>
  struct X {
    S R[9];
  };
  using EpsD = Eps<double>;
  using EpsU = Eps<UD3>;
  static __device__ __forceinline__ EpsU lift(const EpsD& e) { return EpsU{UD3(e.nz), UD3(e.dnz), UD3(e.npi)}; }
  static __device__ __forceinline__ void seed(const double* raw, int k, X<UD3>& y) {
#pragma unroll
    for (int i = 0; i < 9; ++i) y.R[i] = UD3(raw[i], i == k ? 1.0 : 0.0);
  }
  static __device__ __forceinline__ void inv(const X<UD3>& a, X<UD3>& y) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) y.R[3 * i + j] = a.R[3 * j + i];
  }
  static __device__ __forceinline__ void mul(const X<UD3>& a, const X<UD3>& b, X<UD3>& z) { mat3_mul(a.R, b.R, z.R); }
  static __device__ __forceinline__ void adjoint(const X<UD3>& a, UD3* A) {
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = a.R[i];
  }
  static __device__ __forceinline__ void log_jlog(const X<UD3>& E, const EpsU& eps, UD3* xi, UD3* J) {
    so3_log_jlog<UD3>(E.R, eps, xi, J, true);
    // torchlie's log backward: d xi = Jlog vee2(E^T dE) / 2  (vjpso3_kernels.hip follows the same rule)
    double R[9], dR[9], M[9], u[3], jv[9], dxi[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      R[i] = E.R[i].v;
      dR[i] = E.R[i].d;
      jv[i] = J[i].v;
    }
    mat3_tmul(R, dR, M);
    u[0] = 0.5 * (M[7] - M[5]);
    u[1] = 0.5 * (M[2] - M[6]);
    u[2] = 0.5 * (M[3] - M[1]);
    mat3_vec(jv, u, dxi);
#pragma unroll
    for (int i = 0; i < 3; ++i) xi[i].d = dxi[i];
  }
};

// rows of phi and the squared weighted errors of a Between cost (edge = true: Xi, Xj, Z) or a Difference / Local prior (X = Xj,
// T = Z; Xi unused): see unroll_se3.cuh
template <typename G, bool EDGE>
__device__ __forceinline__ void unroll3_phi(const typename G::template X<UD3>& Xi, const typename G::template X<UD3>& Xj,
                                            const typename G::template X<UD3>& Z, const double* s, const double* wi,
                                            const double* wj, const double* di, const double* dj, const typename G::EpsU& eps,
                                            double lam, UD3* phi_r, UD3* x_r, double* a_out, double* b_out, double* ell_out,
                                            double* xi_out) {
  typename G::template X<UD3> Xii, D, Zi, E, Dinv;
  UD3 Ad[9], xi[3], J[9], qw[3], qd[3], a[3], c[3];
  if (EDGE) {
    G::inv(Xi, Xii);
    G::mul(Xii, Xj, D);
    G::inv(D, Dinv);
    G::adjoint(Dinv, Ad);
  } else {
    D = Xj;
  }
  G::inv(Z, Zi);
  G::mul(Zi, D, E);
  G::log_jlog(E, eps, xi, J);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    qw[r] = UD3(wj[r]);
    qd[r] = UD3(dj[r]);
    if (EDGE) {
      qw[r] = qw[r] - (Ad[3 * r] * UD3(wi[0]) + Ad[3 * r + 1] * UD3(wi[1]) + Ad[3 * r + 2] * UD3(wi[2]));
      qd[r] = qd[r] - (Ad[3 * r] * UD3(di[0]) + Ad[3 * r + 1] * UD3(di[1]) + Ad[3 * r + 2] * UD3(di[2]));
    }
  }
  mat3_vec(J, qw, a);
  mat3_vec(J, qd, c);
  UD3 JA[9];
  if (EDGE && lam != 0.0) mat3_mul(J, Ad, JA);   // J_i = -J Ad (the sign drops out of the squares)
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const UD3 bsum = xi[r] + c[r];
    phi_r[r] = UD3(0.0) - UD3(s[r] * s[r]) * (a[r] * bsum);
    x_r[r] = UD3(s[r] * s[r]) * (xi[r] * xi[r]);
    if (a_out) a_out[r] = a[r].v;
    if (b_out) b_out[r] = bsum.v;
    if (xi_out) xi_out[r] = xi[r].v;
    if (lam != 0.0) {   // ellipsoidal damping: -lambda s_r^2 sum_k (J_rk^2 w_jk delta_jk + (J Ad)_rk^2 w_ik delta_ik)
      UD3 e(0.0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        e = e + J[3 * r + k] * J[3 * r + k] * UD3(wj[k] * dj[k]);
        if (EDGE) e = e + JA[3 * r + k] * JA[3 * r + k] * UD3(wi[k] * di[k]);
      }
      phi_r[r] = phi_r[r] - UD3(lam * s[r] * s[r]) * e;
      if (ell_out) ell_out[r] = e.v;
    } else if (ell_out) {
      ell_out[r] = 0.0;
    }
  }
}

// gradients of a cost's Phi = sum_r m_r phi_r: g[0..NR) w.r.t. Xi, [NR..2NR) Xj (the prior's variable), [2NR..3NR) Z (the prior's
// target); gs (3 weights); glr.  raw_*: NR doubles each (raw_i unused for a prior).
template <typename G, bool EDGE>
__device__ __forceinline__ void unroll3_vjp(const double* raw_i, const double* raw_j, const double* raw_z, const double* s,
                                            const double* wi, const double* wj, const double* di, const double* dj,
                                            const typename G::EpsD& eps, double lam, int loss, double log_radius, double* g,
                                            double* gs, double* glr) {
  const typename G::EpsU epsd = G::lift(eps);
  typename G::template X<UD3> A, Bv, C;
  UD3 phi_r[3], x_r[3];
  double a[3], b[3], ell[3], xi[3], P[3], xv[3], pv[3];
  if (EDGE) G::seed(raw_i, -1, A);
  G::seed(raw_j, -1, Bv);
  G::seed(raw_z, -1, C);
  unroll3_phi<G, EDGE>(A, Bv, C, s, wi, wj, di, dj, epsd, lam, phi_r, x_r, a, b, ell, xi);
  RobustTerms<3> rt;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    xv[r] = x_r[r].v;
    pv[r] = phi_r[r].v;
  }
  rt.eval(loss, xv, log_radius);
  rt.group(pv, P);
  double gl = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    gs[r] = rt.m[r] * (-2.0 * s[r] * (a[r] * b[r] + lam * ell[r])) + P[r] * rt.m_x[r] * (2.0 * s[r] * xi[r] * xi[r]);
    gl += pv[r] * rt.m_l[r];
  }
  *glr = gl;
  constexpr int NR = G::NR;
  for (int k = EDGE ? 0 : NR; k < 3 * NR; ++k) {   // run-time loop: one dual evaluation per raw entry
    const int which = k / NR, e = k % NR;
    if (EDGE) G::seed(raw_i, which == 0 ? e : -1, A);
    G::seed(raw_j, which == 1 ? e : -1, Bv);
    G::seed(raw_z, which == 2 ? e : -1, C);
    unroll3_phi<G, EDGE>(A, Bv, C, s, wi, wj, di, dj, epsd, lam, phi_r, x_r, nullptr, nullptr, nullptr, nullptr);
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) acc += rt.m[r] * phi_r[r].d + P[r] * rt.m_x[r] * x_r[r].d;
    g[k] = acc;
  }
}

}  // namespace thx
