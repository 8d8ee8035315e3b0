// SE2 device arithmetic for gfx950, written from the closed forms of theseus/geometry/se2.py (:165-229 log + Jlog,
// :239-300 exp + Jexp, :309-339 adjoint / compose / inverse) and so2.py (:206-234): tensor [x, y, cos, sin],
// tangent [u_x, u_y, theta], right perturbations; Taylor switches keyed by the se2 near_zero / d_near_zero
// thresholds (theseus/global_params.py:46-59) passed in at launch.  Registers only.
#pragma once
#include "lie.cuh"

namespace thx {

template <typename T>
struct Eps2 {
  T nz, dnz;
};
template <typename T>
struct SE2 {
  T x, y, c, s;
};

template <typename T>
__device__ __forceinline__ T t_abs(T v) { return v < T(0) ? -v : v; }

template <typename T>
__device__ __forceinline__ void se2_inv(const SE2<T>& X, SE2<T>& Y) {  // R^-1 = (c, -s); t' = R^-1 (-t)
  Y.x = -(X.c * X.x + X.s * X.y);
  Y.y = -(-X.s * X.x + X.c * X.y);
  Y.c = X.c;
  Y.s = -X.s;
}
template <typename T>
__device__ __forceinline__ void se2_mul(const SE2<T>& A, const SE2<T>& B, SE2<T>& Z) {
  Z.x = A.x + A.c * B.x - A.s * B.y;
  Z.y = A.y + A.s * B.x + A.c * B.y;
  Z.c = A.c * B.c - A.s * B.s;
  Z.s = A.s * B.c + A.c * B.s;
}

// exp (+ optional Jexp, row major 3x3); se2.py:239-300.  NB the small-angle branch of the Jacobian's
// (theta - sin)/theta^2 is literally theta - theta^3/120 in the reference (se2.py:274-276): restated as is.
template <typename T>
__device__ __forceinline__ void se2_exp(const T* xi, const Eps2<T>& eps, SE2<T>& X, T* J) {
  const T ux = xi[0], uy = xi[1], th = xi[2];
  const T cosine = t_cos(th), sine = t_sin(th);
  const bool small = t_abs(th) < eps.nz;
  const T th2 = th * th, th3 = th * th * th;
  const T th_nz = small ? T(1) : th;
  const T sbt = small ? T(1) - th2 / T(6) : sine / th_nz;
  const T cm1bt = small ? -th / T(2) + th3 / T(24) : (cosine - T(1)) / th_nz;
  X.x = sbt * ux + cm1bt * uy;
  X.y = sbt * uy - cm1bt * ux;
  X.c = cosine;
  X.s = sine;
  if (J) {
    const T th2_nz = small ? T(1) : th2;
    const T tms = small ? th - th3 / T(120) : (th - sine) / th2_nz;
    const T cm1bt2 = small ? T(-0.5) + th2 / T(24) : (cosine - T(1)) / th2_nz;
    J[0] = sbt;   J[1] = -cm1bt; J[2] = tms * ux + cm1bt2 * uy;
    J[3] = cm1bt; J[4] = sbt;    J[5] = tms * uy - cm1bt2 * ux;
    J[6] = T(0);  J[7] = T(0);   J[8] = T(1);
  }
}

// log + Jlog (row major 3x3); se2.py:165-229
template <typename T>
__device__ __forceinline__ void se2_log_jlog(const SE2<T>& X, const Eps2<T>& eps, T* xi, T* J, bool want_jac) {
  const T th = t_atan2(X.s, X.c);
  const bool small = t_abs(th) < eps.nz;
  const T sine_nz = small ? T(1) : X.s;
  const T h = T(0.5) * (T(1) + X.c) * (small ? T(1) + X.s * X.s / T(6) : th / sine_nz);
  const T ht = T(0.5) * th;
  const T ux = h * X.x + ht * X.y;
  const T uy = h * X.y - ht * X.x;
  xi[0] = ux; xi[1] = uy; xi[2] = th;
  if (!want_jac) return;
  const T th2 = th * th, th3 = th * th2;
  const bool dsmall = t_abs(th) < eps.dnz;
  const T th_nz = dsmall ? T(1) : th;
  const T omc_nz = dsmall ? T(1) : T(1) - X.c;
  const T a = dsmall ? T(1) - th2 / T(12) : ht * X.s / omc_nz;
  const T k = dsmall ? th / T(12) + th3 / T(720) : T(1) / th_nz - T(0.5) * X.s / omc_nz;
  J[0] = a;    J[1] = -ht;  J[2] = k * ux + T(0.5) * uy;
  J[3] = ht;   J[4] = a;    J[5] = k * uy - T(0.5) * ux;
  J[6] = T(0); J[7] = T(0); J[8] = T(1);
}

// Ad(X) = [[c, -s, y], [s, c, -x], [0, 0, 1]]; se2.py:309-316
template <typename T>
__device__ __forceinline__ void se2_adjoint(const SE2<T>& X, T* A) {
  A[0] = X.c;  A[1] = -X.s; A[2] = X.y;
  A[3] = X.s;  A[4] = X.c;  A[5] = -X.x;
  A[6] = T(0); A[7] = T(0); A[8] = T(1);
}

// Between (embodied/measurements/between.py:38-45) with row weights (core/cost_weight.py:125-136)
template <typename T>
__device__ __forceinline__ void between_eval2(const SE2<T>& v0, const SE2<T>& v1, const SE2<T>& meas, const T* w,
                                              const Eps2<T>& eps, T* e, T* J0, T* J1, bool want_jac) {
  SE2<T> v0i, D, mi, E;
  se2_inv(v0, v0i);
  se2_mul(v0i, v1, D);
  se2_inv(meas, mi);
  se2_mul(mi, D, E);
  T xi[3], Jl[9];
  se2_log_jlog(E, eps, xi, Jl, want_jac);
#pragma unroll
  for (int i = 0; i < 3; ++i) e[i] = xi[i] * w[i];
  if (!want_jac) return;
  SE2<T> Di;
  se2_inv(D, Di);
  T Ad[9], JA[9];
  se2_adjoint(Di, Ad);
  mat3_mul(Jl, Ad, JA);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      J1[3 * i + j] = Jl[3 * i + j] * w[i];
      J0[3 * i + j] = -JA[3 * i + j] * w[i];
    }
}
template <typename T>
__device__ __forceinline__ void local_eval2(const SE2<T>& target, const SE2<T>& var, const T* w, const Eps2<T>& eps,
                                            T* e, T* J, bool want_jac) {
  SE2<T> ti, D;
  se2_inv(target, ti);
  se2_mul(ti, var, D);
  T xi[3], Jl[9];
  se2_log_jlog(D, eps, xi, Jl, want_jac);
#pragma unroll
  for (int i = 0; i < 3; ++i) e[i] = xi[i] * w[i];
  if (!want_jac) return;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) J[3 * i + j] = Jl[3 * i + j] * w[i];
}

// storage <-> fp64 registers (fp32 storage is evaluated in fp64, see lie.cuh "Evaluation precision")
template <typename T>
__device__ __forceinline__ SE2<double> se2_load(const T* __restrict__ p) {
  return SE2<double>{(double)p[0], (double)p[1], (double)p[2], (double)p[3]};
}
template <typename T>
__device__ __forceinline__ void se2_store(T* __restrict__ p, const SE2<double>& X) {
  p[0] = (T)X.x; p[1] = (T)X.y; p[2] = (T)X.c; p[3] = (T)X.s;
}

}  // namespace thx
