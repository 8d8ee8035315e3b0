// Implicit backward for SO3 rotation graphs: the SO3 twin of vjp_kernels.hip, with torchlie's SO3 backward semantics
// (torchlie/torchlie/functional/so3_impl.py): Exp.backward (:336-353) grad_w = Jexp^T vee2(R^T G); Log's passthrough backward
// (:489-496) d log = Jlog vee2(E^T dE) / 2 -- the tangent projection, not the derivative of the closed form; Inverse / Compose
// (:576-577, :702-707) are the plain matrix derivatives; the Jlog closed forms are differentiated by plain autograd (dual numbers
// through so3_log_jlog, Taylor branches included).  vee2(M) = (M21 - M12, M02 - M20, M10 - M01).
//   thx_so3_retract_vjp : grad_delta = step * Jexp(step delta)^T vee2(Y^T G),  Y = X exp(step delta)
//   thx_pgso3_vjp       : grad_theta of phi = w^T g, per cost  phi = - m(x, log_radius) sum_r s_r^2 (Jlog(E) q)_r log(E)_r,
//                         E = Z^T C, q = w_j - D^T w_i (edges, D = X_i^T X_j = C) | w_p (priors), x = |s log E|^2
#include "common.cuh"
#include "dual.cuh"
#include "lie_so3.cuh"
#include "robust.cuh"

namespace thx {

using D2 = Dual<double>;

template <typename T>
__device__ __forceinline__ void load_so3_d(const T* __restrict__ p, double* R) {
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = (double)p[k];
}

// gradient of phi w.r.t. the 9 raw entries of Z (row major), the 3 weights and log_radius
__device__ __forceinline__ void cost_vjp_so3(const double* Z, const double* C, const double* q, const double* s,
                                             const Eps<double>& eps, int loss, double log_radius, double* gZ, double* gs,
                                             double* glr) {
  double E[9], xi[3], J[9], a[3];
  mat3_tmul(Z, C, E);
  so3_log_jlog<double>(E, eps, xi, J, true);
  mat3_vec(J, q, a);
  double phi_r[3], x_r[3], Phi[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    phi_r[r] = -s[r] * s[r] * a[r] * xi[r];
    x_r[r] = (s[r] * xi[r]) * (s[r] * xi[r]);
  }
  RobustTerms<3> rt;   // robust.cuh
  rt.eval(loss, x_r, log_radius);
  rt.group(phi_r, Phi);
  double gl = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    gl += phi_r[r] * rt.m_l[r];
    gs[r] = rt.m[r] * (-2.0 * s[r] * a[r] * xi[r]) + Phi[r] * rt.m_x[r] * (2.0 * s[r] * xi[r] * xi[r]);
  }
  *glr = gl;
  const Eps<D2> epsd{D2(eps.nz), D2(eps.dnz), D2(eps.npi)};
  D2 Cd[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Cd[i] = D2(C[i]);
  for (int k = 0; k < 9; ++k) {  // run-time loop: one dual evaluation per raw entry of Z
    D2 Zd[9], Ed[9], xid[3], Jd[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Zd[i] = D2(Z[i], i == k ? 1.0 : 0.0);
    mat3_tmul(Zd, Cd, Ed);
    so3_log_jlog<D2>(Ed, epsd, xid, Jd, true);
    // torchlie's log backward: d xi = Jlog vee2(E^T dE) / 2
    double dE[9], M[9], u[3], dxi[3], da[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) dE[i] = Ed[i].d;
    mat3_tmul(E, dE, M);
    u[0] = 0.5 * (M[7] - M[5]);
    u[1] = 0.5 * (M[2] - M[6]);
    u[2] = 0.5 * (M[3] - M[1]);
    mat3_vec(J, u, dxi);
#pragma unroll
    for (int i = 0; i < 3; ++i) da[i] = Jd[3 * i].d * q[0] + Jd[3 * i + 1].d * q[1] + Jd[3 * i + 2].d * q[2];
    double g = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
      g += rt.m[r] * (-s[r] * s[r] * (da[r] * xi[r] + a[r] * dxi[r])) + Phi[r] * rt.m_x[r] * (2.0 * s[r] * s[r] * xi[r] * dxi[r]);
    gZ[k] = g;
  }
}

template <typename T>
__global__ void __launch_bounds__(64)
pgso3_vjp_kernel(thx_pg_structure s, thx_pg_data d, const T* __restrict__ wvec, int64_t ldw, T* __restrict__ g_meas,
                 T* __restrict__ g_wb, T* __restrict__ g_tgt, T* __restrict__ g_wp, T* __restrict__ g_lrb, T* __restrict__ g_lrp,
                 Eps<T> eps_t) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int c = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const Eps<double> eps{(double)eps_t.nz, (double)eps_t.dnz, (double)eps_t.npi};
  const T* poses = static_cast<const T*>(d.poses);
  const T* wv = wvec + (int64_t)b * ldw;
  double q[3], sw[3], gZ[9], gs[3], glr = 0.0, lr = 0.0, Z[9], C[9];
  int loss = THX_LOSS_NONE;
  T *outZ, *outS, *outL = nullptr;
  if (c < s.num_edges) {
    const int e = c, i = s.edge_i[e], j = s.edge_j[e];
    const int64_t mB = d.meas_bstride ? B : 1, wB = d.w_between_bstride ? B : 1;
    double Xi[9], Xj[9];
    load_so3_d(poses + ((int64_t)i * B + b) * 9, Xi);
    load_so3_d(poses + ((int64_t)j * B + b) * 9, Xj);
    load_so3_d(static_cast<const T*>(d.meas) + ((int64_t)e * mB) * 9 + (int64_t)b * d.meas_bstride, Z);
    const T* wp = static_cast<const T*>(d.w_between) + ((int64_t)e * wB) * 3 + (int64_t)b * d.w_between_bstride;
    mat3_tmul(Xi, Xj, C);  // D = v0^-1 v1
    double wi[3], Dtw[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) wi[r] = (double)wv[3 * i + r];
    mat3_tvec(C, wi, Dtw);  // Ad(D^-1) w_i = D^T w_i
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      q[r] = (double)wv[3 * j + r] - Dtw[r];
      sw[r] = (double)wp[r];
    }
    outZ = g_meas + ((int64_t)e * B + b) * 9;
    outS = g_wb + ((int64_t)e * B + b) * 3;
    loss = loss_code(d.robust_between, d.loss_between, e);
    if (loss) lr = load_log_radius<T>(d.log_radius_between, e, b, B, d.log_radius_between_bstride);
    if (d.robust_between) outL = g_lrb ? g_lrb + (int64_t)e * B + b : nullptr;   // (a plain cost of a mixed role: 0)
  } else {
    const int k = c - s.num_edges, p = s.prior_pose[k];
    const int64_t tB = d.prior_target_bstride ? B : 1, wB = d.w_prior_bstride ? B : 1;
    load_so3_d(poses + ((int64_t)p * B + b) * 9, C);
    load_so3_d(static_cast<const T*>(d.prior_target) + ((int64_t)k * tB) * 9 + (int64_t)b * d.prior_target_bstride, Z);
    const T* wp = static_cast<const T*>(d.w_prior) + ((int64_t)k * wB) * 3 + (int64_t)b * d.w_prior_bstride;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      q[r] = (double)wv[3 * p + r];
      sw[r] = (double)wp[r];
    }
    outZ = g_tgt + ((int64_t)k * B + b) * 9;
    outS = g_wp + ((int64_t)k * B + b) * 3;
    loss = loss_code(d.robust_prior, d.loss_prior, k);
    if (loss) lr = load_log_radius<T>(d.log_radius_prior, k, b, B, d.log_radius_prior_bstride);
    if (d.robust_prior) outL = g_lrp ? g_lrp + (int64_t)k * B + b : nullptr;
  }
  cost_vjp_so3(Z, C, q, sw, eps, loss, lr, gZ, gs, &glr);
  if (outL) *outL = (T)glr;
#pragma unroll
  for (int k = 0; k < 9; ++k) outZ[k] = (T)gZ[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) outS[k] = (T)gs[k];
}

template <typename T>
__global__ void __launch_bounds__(64)
so3_retract_vjp_kernel(const T* __restrict__ poses, const T* __restrict__ delta, int64_t ldd, T step,
                       const T* __restrict__ gout, T* __restrict__ gdelta, int64_t ldg, int P, int B, Eps<T> eps_t) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B) return;
  const Eps<double> eps{(double)eps_t.nz, (double)eps_t.dnz, (double)eps_t.npi};
  double X[9], G[9], w[3], J[9], Y[9], M[9], u[3];
  load_so3_d(poses + ((int64_t)p * B + b) * 9, X);
  load_so3_d(gout + ((int64_t)p * B + b) * 9, G);
#pragma unroll
  for (int i = 0; i < 3; ++i) w[i] = (double)(delta[(int64_t)b * ldd + 3 * p + i] * step);
  GroupSO3::X Ex;
  GroupSO3::exp(w, eps, Ex, J);
  mat3_mul(X, Ex.R, Y);
  mat3_tmul(Y, G, M);
  u[0] = M[7] - M[5];
  u[1] = M[2] - M[6];
  u[2] = M[3] - M[1];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    gdelta[(int64_t)b * ldg + 3 * p + i] = (T)((J[i] * u[0] + J[3 + i] * u[1] + J[6 + i] * u[2]) * (double)step);
}

}  // namespace thx

using namespace thx;

extern "C" {

int thx_pgso3_vjp(const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw, void* grad_meas,
                  void* grad_w_between, void* grad_prior_target, void* grad_w_prior, void* grad_log_radius_between,
                  void* grad_log_radius_prior, int dtype, const thx_lie_eps* eps, void* stream) {
  if (!s || !d || !w || !eps) return fail("thx_pgso3_vjp: null argument");
  if (s->num_edges > 0 && (!grad_meas || !grad_w_between)) return fail("thx_pgso3_vjp: null edge gradient buffer");
  if (s->num_priors > 0 && (!grad_prior_target || !grad_w_prior)) return fail("thx_pgso3_vjp: null prior gradient buffer");
  if (ldw < 3 * (int64_t)s->num_poses) return fail("thx_pgso3_vjp: ldw < n");
  if (const char* why = check_robust(d)) return fail(why);
  dim3 grid((d->batch + 63) / 64, s->num_edges + s->num_priors), block(64);
  if (grid.y == 0) return 0;
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(pgso3_vjp_kernel<float>, grid, block, 0, as_stream(stream), *s, *d, (const float*)w, ldw,
                                  (float*)grad_meas, (float*)grad_w_between, (float*)grad_prior_target, (float*)grad_w_prior,
                                  (float*)grad_log_radius_between, (float*)grad_log_radius_prior, make_eps<float>(eps)),
               hipLaunchKernelGGL(pgso3_vjp_kernel<double>, grid, block, 0, as_stream(stream), *s, *d, (const double*)w, ldw,
                                  (double*)grad_meas, (double*)grad_w_between, (double*)grad_prior_target,
                                  (double*)grad_w_prior, (double*)grad_log_radius_between, (double*)grad_log_radius_prior,
                                  make_eps<double>(eps)));
  return check_launch("thx_pgso3_vjp");
}

int thx_so3_retract_vjp(const void* poses, const void* delta, int64_t ldd, double step, const void* grad_out,
                        void* grad_delta, int64_t ldg, int32_t P, int32_t B, int dtype, const thx_lie_eps* eps, void* stream) {
  if (!poses || !delta || !grad_out || !grad_delta || !eps || P <= 0 || B <= 0) return fail("bad so3_retract_vjp args");
  dim3 grid((B + 63) / 64, P), block(64);
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(so3_retract_vjp_kernel<float>, grid, block, 0, as_stream(stream), (const float*)poses,
                                  (const float*)delta, ldd, (float)step, (const float*)grad_out, (float*)grad_delta, ldg, P,
                                  B, make_eps<float>(eps)),
               hipLaunchKernelGGL(so3_retract_vjp_kernel<double>, grid, block, 0, as_stream(stream), (const double*)poses,
                                  (const double*)delta, ldd, step, (const double*)grad_out, (double*)grad_delta, ldg, P, B,
                                  make_eps<double>(eps)));
  return check_launch("thx_so3_retract_vjp");
}

}  // extern "C"
