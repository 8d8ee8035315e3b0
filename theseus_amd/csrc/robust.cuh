// RobustCostFunction on device (theseus/core/robust_cost_function.py:87-135) with the losses of
// theseus/core/robust_loss.py:33-62 (Welsch, Huber, Hinge) and :92-113 (Geman-McClure with the GNC control value folded into the radius).  x = squared norm of the WEIGHTED error, log_radius as stored by the reference.
//   linearisation pass:  (J, e) <- sqrt(rho'(x) + 1e-20) (J, e)
//   error metric:        |h|^2 = dim * (rho(x) / dim + 1e-20)
// flatten_dims = True (robust_cost_function.py:89-96,118-133): every residual row is its own term -- x_r = e_r^2, row r of
// (J, e) scaled by sqrt(rho'(x_r) + 1e-20), |h|^2 = sum_r (rho(x_r) + 1e-20).
// A LOSS CODE = THX_LOSS_* | THX_LOSS_FLATTEN; one per cost role, or one per cost (thx_pg_data.loss_between / loss_prior).
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
  if (kind == THX_LOSS_HINGE) return x > r ? 1.0 / (2.0 * sqrt(x) + kLossEps) : 0.0;   // robust_loss.py:60-62
  if (kind == THX_LOSS_GEMAN_MCCLURE) return r * r / ((r + x) * (r + x) + kLossEps);     // robust_loss.py:109-113, r = mu * radius
  return sqrt(r / fmax(x, r) + kLossEps);  // Huber
}
// rho(x)
__device__ __forceinline__ double loss_evaluate(int kind, double x, double log_radius) {
  const double r = exp(log_radius);
  if (kind == THX_LOSS_WELSCH) return r - r * exp(-x / (r + kLossEps));
  if (kind == THX_LOSS_HINGE) return x > r ? sqrt(x) - sqrt(r) : kLossEps;   // robust_loss.py:56-58
  if (kind == THX_LOSS_GEMAN_MCCLURE) return r * x / (r + x + kLossEps);      // robust_loss.py:96-107, r = mu * radius
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
  } else if (kind == THX_LOSS_GEMAN_MCCLURE) {
    // rho' = r^2 / ((r + x)^2 + eps), r = exp(log_radius) = mu * radius
    const double sx = r + x, den = sx * sx + kLossEps;
    m = r * r / den + kRobustEps;
    dm_dx = -2.0 * r * r * sx / (den * den);
    dm_dl = (2.0 * r * den - r * r * 2.0 * sx) / (den * den) * r;
  } else if (kind == THX_LOSS_HINGE) {
    // rho' = 1 / (2 sqrt(x) + eps) beyond the radius, 0 inside: the radius enters through the branch only (no gradient, as in
    // the reference's torch.where)
    if (x > r) {
      const double sx = sqrt(x), den = 2.0 * sx + kLossEps;
      m = 1.0 / den + kRobustEps;
      dm_dx = -1.0 / (den * den * sx);
    } else {
      m = kRobustEps;
      dm_dx = 0.0;
    }
    dm_dl = 0.0;
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
// the loss code of one cost: the per-cost table when the role has one, else the role's code
__device__ __forceinline__ int loss_code(int role_code, const int32_t* __restrict__ per_cost, int64_t entity) {
  return per_cost ? per_cost[entity] : role_code;
}
__host__ __device__ __forceinline__ bool loss_code_valid(int code) {
  const int kind = code & ~THX_LOSS_FLATTEN;
  return code >= 0 && (kind <= THX_LOSS_HINGE || kind == THX_LOSS_GEMAN_MCCLURE) && code != THX_LOSS_FLATTEN;
}
// what the cost contributes to 2 * error_metric
template <int DIM>
__device__ __forceinline__ double robust_sq_error(int code, const double* e, double log_radius) {
  if (code == THX_LOSS_NONE) return sqnorm<DIM>(e);
  const int kind = code & ~THX_LOSS_FLATTEN;
  if (code & THX_LOSS_FLATTEN) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      const double h = sqrt(loss_evaluate(kind, e[i] * e[i], log_radius) + kRobustEps);
      acc += h * h;
    }
    return acc;
  }
  const double h = sqrt(loss_evaluate(kind, sqnorm<DIM>(e), log_radius) / DIM + kRobustEps);
  return DIM * (h * h);
}
// per-row rescale factors f_r = sqrt(rho'(x_r) + eps) (all equal without flatten_dims; 1 for a plain cost)
template <int DIM>
__device__ __forceinline__ void robust_row_scale(int code, const double* e, double log_radius, double* f) {
  if (code == THX_LOSS_NONE) {
#pragma unroll
    for (int i = 0; i < DIM; ++i) f[i] = 1.0;
    return;
  }
  const int kind = code & ~THX_LOSS_FLATTEN;
  if (code & THX_LOSS_FLATTEN) {
#pragma unroll
    for (int i = 0; i < DIM; ++i) f[i] = sqrt(loss_linearize(kind, e[i] * e[i], log_radius) + kRobustEps);
  } else {
    const double s = sqrt(loss_linearize(kind, sqnorm<DIM>(e), log_radius) + kRobustEps);
#pragma unroll
    for (int i = 0; i < DIM; ++i) f[i] = s;
  }
}
// host-side validation of the robust fields of a thx_pg_data (NULL: fine)
inline const char* check_robust(const thx_pg_data* d) {
  if ((d->robust_between && !d->log_radius_between) || (d->robust_prior && !d->log_radius_prior))
    return "robust cost without log_loss_radius";
  if (!loss_code_valid(d->robust_between) || !loss_code_valid(d->robust_prior)) return "bad loss kind";
  if ((d->loss_between && !d->robust_between) || (d->loss_prior && !d->robust_prior))
    return "a per-cost loss table needs a non-zero role code (and log_loss_radius)";
  return nullptr;
}

// Backward terms of a robust cost whose per-row squared weighted errors are x_r: m_r = rho'(X_r) + eps with its partials at
// X_r = x_r (flatten_dims) or X_r = sum x (one term).  With phi = sum_r phi_r the plain cost's scalar, the robust one is
// sum_r m_r phi_r and  d/dtheta = sum_r [ m_r dphi_r + Phi_r m_x,r dx_r ],  d/dlog_radius = sum_r phi_r m_l,r,
// Phi_r = phi_r (flatten_dims) | phi (one term): ``group`` maps the per-row phi_r to Phi_r in place.
template <int DIM>
struct RobustTerms {
  double m[DIM], m_x[DIM], m_l[DIM];
  bool flat;
  __device__ __forceinline__ void eval(int code, const double* x, double log_radius) {
    flat = (code & THX_LOSS_FLATTEN) != 0;
    const int kind = code & ~THX_LOSS_FLATTEN;
    if (code == THX_LOSS_NONE) {
#pragma unroll
      for (int r = 0; r < DIM; ++r) { m[r] = 1.0; m_x[r] = 0.0; m_l[r] = 0.0; }
    } else if (flat) {
#pragma unroll
      for (int r = 0; r < DIM; ++r) rescale2_partials(kind, x[r], log_radius, m[r], m_x[r], m_l[r]);
    } else {
      double xs = 0.0;
#pragma unroll
      for (int r = 0; r < DIM; ++r) xs += x[r];
      rescale2_partials(kind, xs, log_radius, m[0], m_x[0], m_l[0]);
#pragma unroll
      for (int r = 1; r < DIM; ++r) { m[r] = m[0]; m_x[r] = m_x[0]; m_l[r] = m_l[0]; }
    }
  }
  // phi_r -> Phi_r
  __device__ __forceinline__ void group(const double* phi_r, double* Phi) const {
    double tot = 0.0;
#pragma unroll
    for (int r = 0; r < DIM; ++r) tot += phi_r[r];
#pragma unroll
    for (int r = 0; r < DIM; ++r) Phi[r] = flat ? phi_r[r] : tot;
  }
};

}  // namespace thx
