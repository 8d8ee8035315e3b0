// RobustCostFunction on device (theseus/core/robust_cost_function.py:87-135, flatten_dims = False) with the losses of
// theseus/core/robust_loss.py:33-52.  x = squared norm of the WEIGHTED error, log_radius as stored by the reference.
//   linearisation pass:  (J, e) <- sqrt(rho'(x) + 1e-20) (J, e)
//   error metric:        |h|^2 = dim * (rho(x) / dim + 1e-20)
// Evaluated in fp64 registers like the rest of the per-cost chain.
#pragma once
#include "../../include/theseus_hip.h"

namespace thx {

constexpr double kLossEps = 1e-20;    // robust_loss.py:10
constexpr double kRobustEps = 1e-20;  // robust_cost_function.py:52

// rho'(x)
__device__ __forceinline__ double loss_linearize(int kind, double x, double log_radius) {
  const double r = exp(log_radius);
  if (kind == THX_LOSS_WELSCH) return exp(-x / (r + kLossEps));
  return sqrt(r / fmax(x, r) + kLossEps);  // Huber
}
// rho(x)
__device__ __forceinline__ double loss_evaluate(int kind, double x, double log_radius) {
  const double r = exp(log_radius);
  if (kind == THX_LOSS_WELSCH) return r - r * exp(-x / (r + kLossEps));
  return x > r ? 2.0 * sqrt(r * fmax(x, r) + kLossEps) - r : x;  // Huber
}
// m = rho'(x) + 1e-20 (the square of the rescale factor) with its partial derivatives w.r.t. x and log_radius
__device__ __forceinline__ void rescale2_partials(int kind, double x, double log_radius, double& m, double& dm_dx,
                                                  double& dm_dl) {
  const double r = exp(log_radius);
  if (kind == THX_LOSS_WELSCH) {
    const double rr = r + kLossEps, v = exp(-x / rr);
    m = v + kRobustEps;
    dm_dx = -v / rr;
    dm_dl = v * x / (rr * rr) * r;
  } else {
    const double mx = fmax(x, r), v = sqrt(r / mx + kLossEps);
    m = v + kRobustEps;
    if (x > r) {
      dm_dx = -0.5 / v * r / (x * x);
      dm_dl = 0.5 / v * r / x;
    } else {
      dm_dx = 0.0;
      dm_dl = 0.0;
    }
  }
}

template <typename T>
__device__ __forceinline__ double load_log_radius(const void* base, int64_t entity, int b, int B, int64_t bstride) {
  const T* p = static_cast<const T*>(base);
  return (double)p[entity * (bstride ? B : 1) + (int64_t)b * bstride];
}

template <int DIM>
__device__ __forceinline__ double sqnorm(const double* e) {
  double x = 0.0;
#pragma unroll
  for (int i = 0; i < DIM; ++i) x += e[i] * e[i];
  return x;
}
// what the cost contributes to 2 * error_metric
template <int DIM>
__device__ __forceinline__ double robust_sq_error(int kind, const double* e, double log_radius) {
  const double x = sqnorm<DIM>(e);
  if (kind == THX_LOSS_NONE) return x;
  const double h = sqrt(loss_evaluate(kind, x, log_radius) / DIM + kRobustEps);
  return DIM * (h * h);
}
template <int DIM>
__device__ __forceinline__ double robust_rescale(int kind, const double* e, double log_radius) {
  return sqrt(loss_linearize(kind, sqnorm<DIM>(e), log_radius) + kRobustEps);
}

}  // namespace thx
