// Implicit backward (BackwardMode.IMPLICIT, theseus/optimizer/nonlinear/nonlinear_least_squares.py:121-135,
// 265-292): the grad-enabled last step is  X_new = X exp(delta),  delta = H^-1 g(theta)  with H detached
// (dense_linearization.py:61), theta = measurements / prior targets / cost weights.  Backward:
//   1. thx_se3_retract_vjp : grad_delta = Jexp(delta)^T [ Y_R^T G_t ; vee(Y_R^T G_R) ]   (torchlie Exp.backward,
//                            se3_impl.py:313-343, after Compose.backward :739-747)
//   2. thx_chol_solve      : w = H^-1 grad_delta with the cached factor (chol_kernels.hip)
//   3. thx_pg_vjp          : grad_theta = d(w^T g)/d theta, per cost, by forward-mode differentiation
// With q = w_j - Ad(D^-1) w_i (edges; q = w_p for priors) the part of w^T g that belongs to one cost is
//   phi = - sum_r s_r^2 (Jlog(E) q)_r log(E)_r ,   E = Z^-1 C   (Z = measurement, C = v0^-1 v1 | Z = target, C = var)
// d phi / d s_r is closed form; d phi / d Z_k (12 raw entries) uses dual numbers through inverse / compose / the
// Jlog closed forms -- the derivative the reference's plain autograd takes -- while log(E)'s own derivative is
// torchlie's passthrough backward (se3_impl.py:487-493): d log = Jlog [E_R^T dE_t ; vee(E_R^T dE_R)/2].
// RobustCostFunction (robust_cost_function.py:115-135) multiplies the cost's part by m = rho'(x) + 1e-20,
// x = sum_r (s_r log(E)_r)^2, which is NOT detached: d(m phi) = m d phi + phi (m_x dx + m_l d log_radius).
#include "vjp_se3.cuh"

namespace thx {

template <typename T>
__global__ void __launch_bounds__(64)
pg_vjp_kernel(thx_pg_structure s, thx_pg_data d, const T* __restrict__ wvec, int64_t ldw, T* __restrict__ g_meas,
              T* __restrict__ g_wb, T* __restrict__ g_tgt, T* __restrict__ g_wp, T* __restrict__ g_lrb,
              T* __restrict__ g_lrp, Eps<T> eps_t) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int c = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const Eps<double> eps{(double)eps_t.nz, (double)eps_t.dnz, (double)eps_t.npi};
  const T* poses = static_cast<const T*>(d.poses);
  const T* wv = wvec + (int64_t)b * ldw;
  double q[6], sw[6], gZ[12], gs[6], glr = 0.0, lr = 0.0;
  int loss = THX_LOSS_NONE;
  SE3<double> Z, C;
  T* outZ;
  T* outS;
  T* outL = nullptr;
  if (c < s.num_edges) {
    const int e = c, i = s.edge_i[e], j = s.edge_j[e];
    const int64_t mB = d.meas_bstride ? B : 1, wB = d.w_between_bstride ? B : 1;
    SE3<double> Xi, Xj, Xii, Di;
    load_se3_any(poses + ((int64_t)i * B + b) * 12, Xi);
    load_se3_any(poses + ((int64_t)j * B + b) * 12, Xj);
    load_se3_any(static_cast<const T*>(d.meas) + ((int64_t)e * mB) * 12 + (int64_t)b * d.meas_bstride, Z);
    const T* wp = static_cast<const T*>(d.w_between) + ((int64_t)e * wB) * 6 + (int64_t)b * d.w_between_bstride;
    se3_inv(Xi, Xii);
    se3_mul(Xii, Xj, C);  // D = v0^-1 v1
    se3_inv(C, Di);
    // q = w_j - Ad(D^-1) w_i,  Ad = [[R, hat(t) R],[0, R]]
    double wi[6], wj[6], Rl[3], Ra[3], tx[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      wi[k] = (double)wv[6 * i + k];
      wj[k] = (double)wv[6 * j + k];
      sw[k] = (double)wp[k];
    }
    mat3_vec(Di.R, wi, Rl);
    mat3_vec(Di.R, wi + 3, Ra);
    cross3(Di.t, Ra, tx);  // hat(t) (R w_ang)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      q[k] = wj[k] - (Rl[k] + tx[k]);
      q[3 + k] = wj[3 + k] - Ra[k];
    }
    outZ = g_meas + ((int64_t)e * B + b) * 12;
    outS = g_wb + ((int64_t)e * B + b) * 6;
    loss = loss_code(d.robust_between, d.loss_between, e);
    if (loss) lr = load_log_radius<T>(d.log_radius_between, e, b, B, d.log_radius_between_bstride);
    if (d.robust_between) outL = g_lrb ? g_lrb + (int64_t)e * B + b : nullptr;   // (a plain cost of a mixed role: 0)
  } else {
    const int k = c - s.num_edges, p = s.prior_pose[k];
    const int64_t tB = d.prior_target_bstride ? B : 1, wB = d.w_prior_bstride ? B : 1;
    load_se3_any(poses + ((int64_t)p * B + b) * 12, C);
    load_se3_any(static_cast<const T*>(d.prior_target) + ((int64_t)k * tB) * 12 + (int64_t)b * d.prior_target_bstride, Z);
    const T* wp = static_cast<const T*>(d.w_prior) + ((int64_t)k * wB) * 6 + (int64_t)b * d.w_prior_bstride;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      q[r] = (double)wv[6 * p + r];
      sw[r] = (double)wp[r];
    }
    outZ = g_tgt + ((int64_t)k * B + b) * 12;
    outS = g_wp + ((int64_t)k * B + b) * 6;
    loss = loss_code(d.robust_prior, d.loss_prior, k);
    if (loss) lr = load_log_radius<T>(d.log_radius_prior, k, b, B, d.log_radius_prior_bstride);
    if (d.robust_prior) outL = g_lrp ? g_lrp + (int64_t)k * B + b : nullptr;
  }
  cost_vjp(Z, C, q, sw, eps, loss, lr, gZ, gs, &glr);
  if (outL) *outL = (T)glr;
#pragma unroll
  for (int k = 0; k < 12; ++k) outZ[k] = (T)gZ[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) outS[k] = (T)gs[k];
}

template <typename T>
__global__ void __launch_bounds__(64)
se3_retract_vjp_kernel(const T* __restrict__ poses, const T* __restrict__ delta, int64_t ldd, T step,
                       const T* __restrict__ gout, T* __restrict__ gdelta, int64_t ldg, int P, int B, Eps<T> eps_t) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B) return;
  const Eps<double> eps{(double)eps_t.nz, (double)eps_t.dnz, (double)eps_t.npi};
  SE3<double> X, Ex, Y, G;
  load_se3_any(poses + ((int64_t)p * B + b) * 12, X);
  load_se3_any(gout + ((int64_t)p * B + b) * 12, G);
  double xi[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) xi[i] = (double)(delta[(int64_t)b * ldd + 6 * p + i] * step);
  ExpCoef<double> c;
  double Ct, J[36];
  se3_exp(xi, eps, Ex, c, Ct);
  se3_jexp(xi, Ex, c, Ct, J);
  se3_mul(X, Ex, Y);
  double M[9], u[6];
  mat3_tmul(Y.R, G.R, M);
  mat3_tvec(Y.R, G.t, u);
  u[3] = M[7] - M[5];
  u[4] = M[2] - M[6];
  u[5] = M[3] - M[1];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc += J[6 * k + i] * u[k];
    gdelta[(int64_t)b * ldg + 6 * p + i] = (T)(acc * (double)step);
  }
}

}  // namespace thx

using namespace thx;

extern "C" {

int thx_pg_vjp(const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw, void* grad_meas,
               void* grad_w_between, void* grad_prior_target, void* grad_w_prior, void* grad_log_radius_between,
               void* grad_log_radius_prior, int dtype, const thx_lie_eps* eps, void* stream) {
  if (!s || !d || !w || !eps) return fail("thx_pg_vjp: null argument");
  if (s->num_edges > 0 && (!grad_meas || !grad_w_between)) return fail("thx_pg_vjp: null edge gradient buffer");
  if (s->num_priors > 0 && (!grad_prior_target || !grad_w_prior)) return fail("thx_pg_vjp: null prior gradient buffer");
  if (ldw < 6 * (int64_t)s->num_poses) return fail("thx_pg_vjp: ldw < n");
  if (const char* why = check_robust(d)) return fail(why);
  dim3 grid((d->batch + 63) / 64, s->num_edges + s->num_priors), block(64);
  if (grid.y == 0) return 0;
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(pg_vjp_kernel<float>, grid, block, 0, as_stream(stream), *s, *d, (const float*)w, ldw,
                                  (float*)grad_meas, (float*)grad_w_between, (float*)grad_prior_target,
                                  (float*)grad_w_prior, (float*)grad_log_radius_between, (float*)grad_log_radius_prior,
                                  make_eps<float>(eps)),
               hipLaunchKernelGGL(pg_vjp_kernel<double>, grid, block, 0, as_stream(stream), *s, *d, (const double*)w,
                                  ldw, (double*)grad_meas, (double*)grad_w_between, (double*)grad_prior_target,
                                  (double*)grad_w_prior, (double*)grad_log_radius_between,
                                  (double*)grad_log_radius_prior, make_eps<double>(eps)));
  return check_launch("thx_pg_vjp");
}

int thx_se3_retract_vjp(const void* poses, const void* delta, int64_t ldd, double step, const void* grad_out,
                        void* grad_delta, int64_t ldg, int32_t P, int32_t B, int dtype, const thx_lie_eps* eps,
                        void* stream) {
  if (!poses || !delta || !grad_out || !grad_delta || !eps || P <= 0 || B <= 0) return fail("bad retract_vjp args");
  dim3 grid((B + 63) / 64, P), block(64);
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(se3_retract_vjp_kernel<float>, grid, block, 0, as_stream(stream), (const float*)poses,
                                  (const float*)delta, ldd, (float)step, (const float*)grad_out, (float*)grad_delta,
                                  ldg, P, B, make_eps<float>(eps)),
               hipLaunchKernelGGL(se3_retract_vjp_kernel<double>, grid, block, 0, as_stream(stream),
                                  (const double*)poses, (const double*)delta, ldd, step, (const double*)grad_out,
                                  (double*)grad_delta, ldg, P, B, make_eps<double>(eps)));
  return check_launch("thx_se3_retract_vjp");
}

}  // extern "C"
