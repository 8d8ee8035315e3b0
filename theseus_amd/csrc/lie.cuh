// SE3 / SO3 device arithmetic for gfx950, written from the maths in SURVEY.md Appendix A.
// Behaviour (not code) follows torchlie: tangent order [v, w], right perturbations, Taylor
// switches keyed by near_zero / d_near_zero / near_pi thresholds passed in at launch
// (reference: torchlie/torchlie/functional/so3_impl.py:220-261,390-479,
//  torchlie/torchlie/functional/se3_impl.py:178-216,354-457,531-538,578-581,703-708).
// Everything lives in registers; no dynamic indexing (would go to scratch on CDNA).
#pragma once
#include <hip/hip_runtime.h>

namespace thx {

template <typename T>
struct Eps {
  T nz, dnz, npi;
};

__device__ __forceinline__ float t_sin(float x) { return sinf(x); }
__device__ __forceinline__ double t_sin(double x) { return sin(x); }
__device__ __forceinline__ float t_cos(float x) { return cosf(x); }
__device__ __forceinline__ double t_cos(double x) { return cos(x); }
__device__ __forceinline__ float t_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double t_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float t_atan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double t_atan2(double y, double x) { return atan2(y, x); }

template <typename T>
struct SE3 {
  T R[9];  // row major
  T t[3];
};

// 3x3 helpers (row major, fully unrolled) -------------------------------------------------------
template <typename T>
__device__ __forceinline__ void mat3_mul(const T* A, const T* B, T* C) {  // C = A B
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
template <typename T>
__device__ __forceinline__ void mat3_tmul(const T* A, const T* B, T* C) {  // C = A^T B
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
template <typename T>
__device__ __forceinline__ void mat3_mult(const T* A, const T* B, T* C) {  // C = A B^T
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
template <typename T>
__device__ __forceinline__ void mat3_vec(const T* A, const T* v, T* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
template <typename T>
__device__ __forceinline__ void mat3_tvec(const T* A, const T* v, T* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}
template <typename T>
__device__ __forceinline__ void cross3(const T* a, const T* b, T* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
// M += hat(s*w): hat(w)[0,1]=-w2, [0,2]=w1, [1,2]=-w0
template <typename T>
__device__ __forceinline__ void add_hat(T* M, const T* w, T s) {
  M[1] -= s * w[2];
  M[3] += s * w[2];
  M[2] += s * w[1];
  M[6] -= s * w[1];
  M[5] -= s * w[0];
  M[7] += s * w[0];
}

// group ops ---------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void se3_load(const T* __restrict__ p, SE3<T>& X) {  // (3,4) row major
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    X.R[3 * i] = p[4 * i];
    X.R[3 * i + 1] = p[4 * i + 1];
    X.R[3 * i + 2] = p[4 * i + 2];
    X.t[i] = p[4 * i + 3];
  }
}
template <typename T>
__device__ __forceinline__ void se3_store(T* __restrict__ p, const SE3<T>& X) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    p[4 * i] = X.R[3 * i];
    p[4 * i + 1] = X.R[3 * i + 1];
    p[4 * i + 2] = X.R[3 * i + 2];
    p[4 * i + 3] = X.t[i];
  }
}
template <typename T>
__device__ __forceinline__ void se3_inv(const SE3<T>& X, SE3<T>& Y) {  // [R^T | -R^T t]
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Y.R[3 * i + j] = X.R[3 * j + i];
  T q[3];
  mat3_vec(Y.R, X.t, q);
  Y.t[0] = -q[0];
  Y.t[1] = -q[1];
  Y.t[2] = -q[2];
}
template <typename T>
__device__ __forceinline__ void se3_mul(const SE3<T>& X, const SE3<T>& Y, SE3<T>& Z) {  // [R0R1 | R0 t1 + t0]
  mat3_mul(X.R, Y.R, Z.R);
  T q[3];
  mat3_vec(X.R, Y.t, q);
  Z.t[0] = q[0] + X.t[0];
  Z.t[1] = q[1] + X.t[1];
  Z.t[2] = q[2] + X.t[2];
}

// SO3 exp with the coefficients SE3 needs (so3_impl.py:220-261)
template <typename T>
struct ExpCoef {
  T theta, theta2, theta_nz, theta2_nz, sine, cosine, A, B;
  bool nz;
};
template <typename T>
__device__ __forceinline__ void so3_exp(const T* w, const Eps<T>& eps, T* R, ExpCoef<T>& c) {
  c.theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  c.theta = t_sqrt(c.theta2);
  c.theta2 = c.theta * c.theta;  // reference squares the norm
  c.nz = c.theta < eps.nz;
  c.theta_nz = c.nz ? T(1) : c.theta;
  c.theta2_nz = c.nz ? T(1) : c.theta2;
  c.cosine = c.nz ? T(8) / (T(4) + c.theta2) - T(1) : t_cos(c.theta);
  c.sine = t_sin(c.theta);
  c.A = c.nz ? T(0.5) * c.cosine + T(0.5) : c.sine / c.theta_nz;
  c.B = c.nz ? T(0.5) * c.A : (T(1) - c.cosine) / c.theta2_nz;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[3 * i + j] = c.B * w[i] * w[j];
  R[0] += c.cosine;
  R[4] += c.cosine;
  R[8] += c.cosine;
  add_hat(R, w, c.A);
}

// SE3 exp (se3_impl.py:178-216); xi = [v, w]
template <typename T>
__device__ __forceinline__ void se3_exp(const T* xi, const Eps<T>& eps, SE3<T>& X, ExpCoef<T>& c, T& Ct) {
  const T* v = xi;
  const T* w = xi + 3;
  so3_exp(w, eps, X.R, c);
  T theta3_nz = c.theta_nz * c.theta2_nz;
  Ct = c.nz ? T(1) / T(6) - c.theta2 / T(120) : (c.theta - c.sine) / theta3_nz;
  T wxv[3];
  cross3(w, v, wxv);
  T wv = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) X.t[i] = c.A * v[i] + c.B * wxv[i] + Ct * (w[i] * wv);
}

// SE3 Jexp (se3_impl.py:225-310): J = [[Jr, R^T Q],[0, Jr]] as full row-major 6x6
template <typename T>
__device__ __forceinline__ void se3_jexp(const T* xi, const SE3<T>& X, const ExpCoef<T>& c, T Ct, T* J) {
  const T* v = xi;
  const T* w = xi + 3;
  T Crot = c.nz ? T(0) : Ct;
  T Jr[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Jr[3 * i + j] = Crot * w[i] * w[j];
  Jr[0] += c.A;
  Jr[4] += c.A;
  Jr[8] += c.A;
  add_hat(Jr, w, -c.B);
  T dB = c.nz ? T(-1) / T(12) : (c.A - T(2) * c.B) / c.theta2_nz;
  T dC = c.nz ? T(-1) / T(60) : (c.B - T(3) * Ct) / c.theta2_nz;
  T wv[3], wwv[3], sw[3], a[3], tv[3];
  cross3(w, v, wv);
  cross3(w, wv, wwv);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    sw[i] = Ct * w[i];
    a[i] = dB * wv[i] + dC * wwv[i];
    tv[i] = -c.B * v[i] - Ct * wv[i];
  }
  T Q[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Q[3 * i + j] = a[i] * w[j] - v[i] * sw[j];
  add_hat(Q, tv, T(1));
  T swv = sw[0] * v[0] + sw[1] * v[1] + sw[2] * v[2];
  Q[0] += swv;
  Q[4] += swv;
  Q[8] += swv;
  T RtQ[9];
  mat3_tmul(X.R, Q, RtQ);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      J[6 * i + j] = Jr[3 * i + j];
      J[6 * i + 3 + j] = RtQ[3 * i + j];
      J[6 * (i + 3) + j] = T(0);
      J[6 * (i + 3) + 3 + j] = Jr[3 * i + j];
    }
}

// SO3 log (so3_impl.py:366-433): w, theta, sine, cosine
template <typename T>
__device__ __forceinline__ void so3_log(const T* R, const Eps<T>& eps, T* w, T& theta, T& sine, T& cosine) {
  T sa[3];
  sa[0] = T(0.5) * (R[7] - R[5]);
  sa[1] = T(0.5) * (R[2] - R[6]);
  sa[2] = T(0.5) * (R[3] - R[1]);
  cosine = T(0.5) * (R[0] + R[4] + R[8] - T(1));
  sine = t_sqrt(sa[0] * sa[0] + sa[1] * sa[1] + sa[2] * sa[2]);
  theta = t_atan2(sine, cosine);
  bool nz = theta < eps.nz;
  bool npi = (T(1) + cosine) <= eps.npi;
  bool nznp = nz || npi;
  T sine_nz = nznp ? T(1) : sine;
  T scale = nznp ? T(1) + sine * sine / T(6) : theta / sine_nz;
  // near pi: major diagonal axis extraction
  T d0 = R[0], d1 = R[4], d2 = R[8];
  bool m1 = (d1 > d0) && (d1 > d2);
  bool m2 = (d2 > d0) && (d2 > d1);
  int major = (m1 ? 1 : 0) + (m2 ? 2 : 0);  // 0,1,2 (3 impossible)
  T sel[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    T rowv = major == 0 ? R[k] : (major == 1 ? R[3 + k] : R[6 + k]);        // R[major][k]
    T colv = major == 0 ? R[3 * k] : (major == 1 ? R[3 * k + 1] : R[3 * k + 2]);  // R[k][major]
    sel[k] = T(0.5) * (rowv + colv);
    sel[k] -= (k == major) ? cosine : T(0);
  }
  T nrm = nz ? T(1) : t_sqrt(sel[0] * sel[0] + sel[1] * sel[1] + sel[2] * sel[2]);
  T sam = major == 0 ? sa[0] : (major == 1 ? sa[1] : sa[2]);
  T sgn = sam > T(0) ? T(1) : (sam < T(0) ? T(-1) : T(1));
  T ts = theta * sgn;
#pragma unroll
  for (int k = 0; k < 3; ++k) w[k] = npi ? (sel[k] / nrm) * ts : sa[k] * scale;
}

// Structured SE3 log + Jlog (se3_impl.py:354-457).  Jlog = [[Jr, Jt],[0, Jr]].
template <typename T>
__device__ __forceinline__ void se3_log_jlog(const SE3<T>& X, const Eps<T>& eps, T* xi, T* Jr, T* Jt,
                                             bool want_jac) {
  T w[3], theta, sine, cosine;
  so3_log(X.R, eps, w, theta, sine, cosine);
  bool nz = theta < eps.nz;
  T theta2 = theta * theta;
  T st = sine * theta;
  T tcm2 = T(2) * cosine - T(2);
  T tcm2_nz = nz ? T(1) : tcm2;
  T theta2_nz = nz ? T(1) : theta2;
  T a = nz ? T(1) - theta2 / T(12) : -st / tcm2_nz;
  T b = nz ? T(1) / T(12) + theta2 / T(720) : (st + tcm2) / (theta2_nz * tcm2_nz);
  T wxt[3];
  cross3(w, X.t, wxt);
  T wt = w[0] * X.t[0] + w[1] * X.t[1] + w[2] * X.t[2];
  T v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    v[i] = a * X.t[i] - T(0.5) * wxt[i] + b * (w[i] * wt);
    xi[i] = v[i];
    xi[3 + i] = w[i];
  }
  if (!want_jac) return;
  // SO3 Jlog with d_near_zero (so3_impl.py:442-479)
  bool dnz = theta < eps.dnz;
  T tcm2_d = dnz ? T(1) : tcm2;
  T theta2_d = dnz ? T(1) : theta2;
  T a2 = dnz ? T(1) - theta2 / T(12) : -st / tcm2_d;
  T b2 = dnz ? T(1) / T(12) + theta2 / T(720) : (st + tcm2) / (theta2_d * tcm2_d);
  T bw[3] = {b2 * w[0], b2 * w[1], b2 * w[2]};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Jr[3 * i + j] = bw[i] * w[j];
  add_hat(Jr, w, T(0.5));
  Jr[0] += a2;
  Jr[4] += a2;
  Jr[8] += a2;
  // translation block (se3_impl.py:419-455); theta2_nz / tcm2_nz come from the near_zero switch
  T theta_d = dnz ? T(1) : theta;
  T theta4_nz = theta2_nz * theta2_nz;
  T cc = dnz ? T(-1) / T(360) - theta2 / T(7560)
             : -(T(2) * tcm2_nz + theta * sine + theta2) / (theta4_nz * tcm2_nz);
  T dd = dnz ? T(-1) / T(6) - theta2 / T(180) : (theta - sine) / (theta_d * tcm2_nz);
  T e = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
  T ce = cc * e;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Jt[3 * i + j] = (ce * w[i]) * w[j] + (bw[i] * v[j] + v[i] * bw[j]);
  T ed = e * dd;
  Jt[0] += ed;
  Jt[4] += ed;
  Jt[8] += ed;
  add_hat(Jt, v, T(0.5));
}

// Weighted structured 6x6 Jacobian  W*[[TL, TR],[0, BR]]  kept as three 3x3 blocks.
template <typename T>
struct SJac {
  T a[9];  // diag(w[0:3]) * TL
  T c[9];  // diag(w[0:3]) * TR
  T d[9];  // diag(w[3:6]) * BR
};

// Between residual + Jacobians (theseus/embodied/measurements/between.py:38-45) with
// DiagonalCostWeight row scaling (theseus/core/cost_weight.py:125-136):
//   D = v0^-1 v1 ; E = m^-1 D ; e = log E ; J1 = Jlog(E) ; J0 = -J1 Ad(D^-1)
// Ad(D^-1) = [[Rt, hat(t')Rt],[0, Rt]] with D^-1 = [Rt | t'].
template <typename T>
__device__ __forceinline__ void between_eval(const SE3<T>& v0, const SE3<T>& v1, const SE3<T>& meas,
                                             const T* w, const Eps<T>& eps, T* e, SJac<T>* J0,
                                             SJac<T>* J1, bool want_jac) {
  SE3<T> v0i, D, mi, E;
  se3_inv(v0, v0i);
  se3_mul(v0i, v1, D);
  se3_inv(meas, mi);
  se3_mul(mi, D, E);
  T xi[6], Jr[9], Jt[9];
  se3_log_jlog(E, eps, xi, Jr, Jt, want_jac);
#pragma unroll
  for (int i = 0; i < 6; ++i) e[i] = xi[i] * w[i];
  if (!want_jac) return;
  SE3<T> Di;
  se3_inv(D, Di);
  // hat(t') Rt
  T H[9] = {T(0), -Di.t[2], Di.t[1], Di.t[2], T(0), -Di.t[0], -Di.t[1], Di.t[0], T(0)};
  T HR[9], JrR[9], JrHR[9], JtR[9];
  mat3_mul(H, Di.R, HR);
  mat3_mul(Jr, Di.R, JrR);
  mat3_mul(Jr, HR, JrHR);
  mat3_mul(Jt, Di.R, JtR);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int k = 3 * i + j;
      J1->a[k] = Jr[k] * w[i];
      J1->c[k] = Jt[k] * w[i];
      J1->d[k] = Jr[k] * w[3 + i];
      J0->a[k] = (-JrR[k]) * w[i];
      J0->c[k] = (-(JrHR[k] + JtR[k])) * w[i];
      J0->d[k] = (-JrR[k]) * w[3 + i];
    }
}

// Local / Difference cost (theseus/embodied/misc/local_cost_fn.py:58-61, lie_group.py:180-195):
//   e = log(target^-1 var), J = Jlog
template <typename T>
__device__ __forceinline__ void local_eval(const SE3<T>& target, const SE3<T>& var, const T* w,
                                           const Eps<T>& eps, T* e, SJac<T>* J, bool want_jac) {
  SE3<T> ti, D;
  se3_inv(target, ti);
  se3_mul(ti, var, D);
  T xi[6], Jr[9], Jt[9];
  se3_log_jlog(D, eps, xi, Jr, Jt, want_jac);
#pragma unroll
  for (int i = 0; i < 6; ++i) e[i] = xi[i] * w[i];
  if (!want_jac) return;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int k = 3 * i + j;
      J->a[k] = Jr[k] * w[i];
      J->c[k] = Jt[k] * w[i];
      J->d[k] = Jr[k] * w[3 + i];
    }
}

// Expand a structured Jacobian to a dense row-major 6x6
template <typename T>
__device__ __forceinline__ void sjac_dense(const SJac<T>& J, T* M) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      M[6 * i + j] = J.a[3 * i + j];
      M[6 * i + 3 + j] = J.c[3 * i + j];
      M[6 * (i + 3) + j] = T(0);
      M[6 * (i + 3) + 3 + j] = J.d[3 * i + j];
    }
}

// Blk(6x6 row major) += P^T Q for structured P, Q:
//   [[Pa^T Qa, Pa^T Qc],[Pc^T Qa, Pc^T Qc + Pd^T Qd]]
// (C = A^T B with an EXPLICIT fma chain: the Hessian blocks are accumulated by several instantiations of the assembly kernel --
//  dense frame / block list -- whose values must agree bit for bit, and the compiler's contraction of a*b + c*d + e*f depends
//  on the surrounding code)
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T>
__device__ __forceinline__ void mat3_tmul_fma(const T* A, const T* B, T* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = fma_t(A[6 + i], B[6 + j], fma_t(A[3 + i], B[3 + j], A[i] * B[j]));
}
template <typename T>
__device__ __forceinline__ void sjac_tmul_acc(const SJac<T>& P, const SJac<T>& Q, T* Blk) {
#define mat3_tmul mat3_tmul_fma
  T m[9];
  mat3_tmul(P.a, Q.a, m);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Blk[6 * i + j] += m[3 * i + j];
  mat3_tmul(P.a, Q.c, m);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Blk[6 * i + 3 + j] += m[3 * i + j];
  mat3_tmul(P.c, Q.a, m);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Blk[6 * (i + 3) + j] += m[3 * i + j];
  T m2[9];
  mat3_tmul(P.c, Q.c, m);
  mat3_tmul(P.d, Q.d, m2);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Blk[6 * (i + 3) + 3 + j] += m[3 * i + j] + m2[3 * i + j];
#undef mat3_tmul
}

// g(6) -= P^T e     (Atb = A^T b with b = -err, theseus/optimizer/dense_linearization.py:55)
template <typename T>
__device__ __forceinline__ void sjac_tvec_sub(const SJac<T>& P, const T* e, T* g) {
  T q[3], r[3], s[3];
  mat3_tvec(P.a, e, q);
  mat3_tvec(P.c, e, r);
  mat3_tvec(P.d, e + 3, s);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g[i] -= q[i];
    g[3 + i] -= r[i] + s[i];
  }
}

// ------------------------------------------------------------------------------------------------
// Evaluation precision.  The fp32 path keeps fp32 STORAGE (poses, measurements, H, g, L) and fp32
// MFMA for the dense factorisation, but evaluates the per-edge Lie chain (compose, log, Jlog, Ad)
// in fp64 REGISTERS: the reference's closed forms cancel catastrophically in fp32 for the small
// residual rotations a pose graph produces (a = -s*th/(2c-2), b = (s*th+2c-2)/(th^2 (2c-2)) with
// th ~ 1e-2..1e-1: 2c-2 ~ -th^2 carries an absolute error of ~1e-7, so b is wrong by O(1)), which
// makes any fp32 evaluation -- the reference's included -- a draw from a noise band of ~1e-4..1e-3
// relative on the translation residual.  MI355X runs vector fp64 at half the fp32 rate and the
// assembly is HBM bound, so we evaluate at the centre of that band instead of adding another draw.
// Thresholds stay the float-rounded ones the reference compares against.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ SE3<double> widen(const SE3<T>& X) {
  SE3<double> Y;
#pragma unroll
  for (int i = 0; i < 9; ++i) Y.R[i] = (double)X.R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) Y.t[i] = (double)X.t[i];
  return Y;
}
template <typename T>
__device__ __forceinline__ SE3<T> narrow(const SE3<double>& X) {
  SE3<T> Y;
#pragma unroll
  for (int i = 0; i < 9; ++i) Y.R[i] = (T)X.R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) Y.t[i] = (T)X.t[i];
  return Y;
}
template <typename T>
__device__ __forceinline__ Eps<double> widen(const Eps<T>& e) {
  return Eps<double>{(double)e.nz, (double)e.dnz, (double)e.npi};
}
template <typename T>
__device__ __forceinline__ SJac<T> narrow(const SJac<double>& J) {
  SJac<T> K;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    K.a[i] = (T)J.a[i];
    K.c[i] = (T)J.c[i];
    K.d[i] = (T)J.d[i];
  }
  return K;
}

// Between / Local evaluated in fp64 registers from T-typed storage; e and Jacobians stay fp64 so the
// caller can accumulate g = -J^T e without a second rounding.
template <typename T>
__device__ __forceinline__ void between_eval_hp(const SE3<T>& v0, const SE3<T>& v1, const SE3<T>& meas,
                                                const T* w, const Eps<T>& eps, double* e, SJac<double>* J0,
                                                SJac<double>* J1, bool want_jac) {
  double wd[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) wd[i] = (double)w[i];
  between_eval<double>(widen(v0), widen(v1), widen(meas), wd, widen(eps), e, J0, J1, want_jac);
}
template <typename T>
__device__ __forceinline__ void local_eval_hp(const SE3<T>& target, const SE3<T>& var, const T* w,
                                              const Eps<T>& eps, double* e, SJac<double>* J, bool want_jac) {
  double wd[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) wd[i] = (double)w[i];
  local_eval<double>(widen(target), widen(var), wd, widen(eps), e, J, want_jac);
}

}  // namespace thx
