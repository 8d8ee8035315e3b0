// SO3 device arithmetic for gfx950 as a group functor for pg3_generic.cuh, written from the closed forms of torchlie
// (torchlie/torchlie/functional/so3_impl.py:220-261 exp, :270-320 Jexp, :390-433 log, :442-479 Jlog, adjoint = R,
// inverse = R^T, compose = R0 R1; theseus/geometry/so3.py wraps exactly these): tensor (3,3) row major, tangent w (3),
// right perturbations, Taylor switches keyed by the so3 near_zero / d_near_zero / near_pi thresholds passed in at launch.
// The building blocks (so3_exp with its coefficients, so3_log with the near-pi axis extraction) are the ones SE3 uses
// (lie.cuh).  Registers only.
#pragma once
#include "lie.cuh"

namespace thx {

template <typename T>
struct SO3m {
  T R[9];
};

// log (+ Jlog = b w w^T + a I + hat(w)/2, sine / cosine taken from the matrix, d_near_zero switch): so3_impl.py:390-479.
// On any scalar type: the implicit backward evaluates it on dual numbers (vjpso3_kernels.hip).
template <typename S>
__device__ __forceinline__ void so3_log_jlog(const S* R, const thx::Eps<S>& eps, S* w, S* J, bool want_jac) {
  S theta, sine, cosine;
  so3_log(R, eps, w, theta, sine, cosine);
  if (!want_jac) return;
  const bool dnz = theta < eps.dnz;
  const S theta2 = theta * theta, st = sine * theta, tcm2 = S(2.0) * cosine - S(2.0);
  const S tcm2_d = dnz ? S(1.0) : tcm2, theta2_d = dnz ? S(1.0) : theta2;
  const S a = dnz ? S(1.0) - theta2 / S(12.0) : -st / tcm2_d;
  const S b = dnz ? S(1.0 / 12.0) + theta2 / S(720.0) : (st + tcm2) / (theta2_d * tcm2_d);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) J[3 * i + j] = (b * w[i]) * w[j];
  add_hat(J, w, S(0.5));
  J[0] += a;
  J[4] += a;
  J[8] += a;
}

struct GroupSO3 {
  static constexpr int REC = 9, DOF = 3;
  using X = SO3m<double>;
  using Eps = thx::Eps<double>;

  template <typename T>
  static __device__ __forceinline__ X load(const T* __restrict__ p) {
    X x;
#pragma unroll
    for (int k = 0; k < 9; ++k) x.R[k] = (double)p[k];
    return x;
  }
  template <typename T>
  static __device__ __forceinline__ void store(T* __restrict__ p, const X& x) {
#pragma unroll
    for (int k = 0; k < 9; ++k) p[k] = (T)x.R[k];
  }
  static __device__ __forceinline__ void inv(const X& a, X& y) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) y.R[3 * i + j] = a.R[3 * j + i];
  }
  static __device__ __forceinline__ void mul(const X& a, const X& b, X& z) { mat3_mul(a.R, b.R, z.R); }
  // exp (+ Jexp = A I - hat(B w) + C w w^T with C = 0 near zero): so3_impl.py:220-261,270-320
  static __device__ __forceinline__ void exp(const double* w, const Eps& eps, X& x, double* J) {
    ExpCoef<double> c;
    so3_exp(w, eps, x.R, c);
    if (!J) return;
    const double theta3_nz = c.theta_nz * c.theta2_nz;
    const double C = c.nz ? 0.0 : (c.theta - c.sine) / theta3_nz;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) J[3 * i + j] = C * w[i] * w[j];
    J[0] += c.A;
    J[4] += c.A;
    J[8] += c.A;
    add_hat(J, w, -c.B);
  }
  // log (+ Jlog): so3_log_jlog below on doubles
  static __device__ __forceinline__ void log_jlog(const X& x, const Eps& eps, double* w, double* J, bool want_jac);
  static __device__ __forceinline__ void adjoint(const X& x, double* A) {
#pragma unroll
    for (int k = 0; k < 9; ++k) A[k] = x.R[k];
  }
};

__device__ __forceinline__ void GroupSO3::log_jlog(const X& x, const Eps& eps, double* w, double* J, bool want_jac) {
  so3_log_jlog<double>(x.R, eps, w, J, want_jac);
}

}  // namespace thx
