// Forward-mode dual numbers for the implicit-backward VJP kernel: the templated Lie arithmetic of lie.cuh is
// instantiated on Dual<double>, which differentiates the SAME closed forms (Taylor branches included) that the
// reference's autograd differentiates (torchlie's jlog / inverse / compose have plain autograd graphs).
#pragma once
#include "lie.cuh"

namespace thx {

template <typename T>
struct Dual {
  T v, d;
  __device__ __forceinline__ Dual() = default;
  __device__ __forceinline__ Dual(T v_) : v(v_), d(T(0)) {}
  __device__ __forceinline__ Dual(T v_, T d_) : v(v_), d(d_) {}
};

#define THX_DUAL_OP __device__ __forceinline__
template <typename T> THX_DUAL_OP Dual<T> operator+(Dual<T> a, Dual<T> b) { return {a.v + b.v, a.d + b.d}; }
template <typename T> THX_DUAL_OP Dual<T> operator-(Dual<T> a, Dual<T> b) { return {a.v - b.v, a.d - b.d}; }
template <typename T> THX_DUAL_OP Dual<T> operator-(Dual<T> a) { return {-a.v, -a.d}; }
template <typename T> THX_DUAL_OP Dual<T> operator*(Dual<T> a, Dual<T> b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
template <typename T> THX_DUAL_OP Dual<T> operator/(Dual<T> a, Dual<T> b) {
  const T q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
template <typename T> THX_DUAL_OP Dual<T>& operator+=(Dual<T>& a, Dual<T> b) { a = a + b; return a; }
template <typename T> THX_DUAL_OP Dual<T>& operator-=(Dual<T>& a, Dual<T> b) { a = a - b; return a; }
template <typename T> THX_DUAL_OP bool operator<(Dual<T> a, Dual<T> b) { return a.v < b.v; }
template <typename T> THX_DUAL_OP bool operator>(Dual<T> a, Dual<T> b) { return a.v > b.v; }
template <typename T> THX_DUAL_OP bool operator<=(Dual<T> a, Dual<T> b) { return a.v <= b.v; }
template <typename T> THX_DUAL_OP bool operator>=(Dual<T> a, Dual<T> b) { return a.v >= b.v; }

template <typename T> THX_DUAL_OP Dual<T> t_sin(Dual<T> x) { return {t_sin(x.v), t_cos(x.v) * x.d}; }
template <typename T> THX_DUAL_OP Dual<T> t_cos(Dual<T> x) { return {t_cos(x.v), -t_sin(x.v) * x.d}; }
// d sqrt at 0 := 0, as torch.linalg.norm's backward does (the reference takes norms of the sine axis)
template <typename T> THX_DUAL_OP Dual<T> t_sqrt(Dual<T> x) {
  const T s = t_sqrt(x.v);
  return {s, x.v > T(0) ? x.d / (T(2) * s) : T(0)};
}
template <typename T> THX_DUAL_OP Dual<T> t_atan2(Dual<T> y, Dual<T> x) {
  const T r2 = x.v * x.v + y.v * y.v;
  return {t_atan2(y.v, x.v), r2 > T(0) ? (x.v * y.d - y.v * x.d) / r2 : T(0)};
}
#undef THX_DUAL_OP

}  // namespace thx
