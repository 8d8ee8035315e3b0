// Implicit backward for SE2 pose graphs: the SE2 twin of vjp_kernels.hip.  theseus/geometry/se2.py has no custom backward
// anywhere (plain autograd through the closed forms, atan2 included), so every derivative is the dual part of the SAME
// templated arithmetic (lie_se2.cuh on Dual<double>, Taylor branches included):
//   thx_se2_retract_vjp : grad_delta_k = < grad_X_new , d/d delta_k [ X exp(step * delta) ] >
//   thx_pg2_vjp         : grad_theta of phi = w^T g, per cost  phi = - m(x, log_radius) sum_r s_r^2 (Jlog(E) q)_r log(E)_r,
//                         E = Z^-1 C, q = w_j - Ad(D^-1) w_i (edges) | w_p (priors), x = |s log E|^2, m = robust rescale^2
#include "common.cuh"
#include "dual.cuh"
#include "lie_se2.cuh"
#include "robust.cuh"

namespace thx {

using D2 = Dual<double>;

template <typename T>
__device__ __forceinline__ SE2<double> se2_load_d(const T* __restrict__ p) {
  return SE2<double>{(double)p[0], (double)p[1], (double)p[2], (double)p[3]};
}

// per row r: phi_r = - s_r^2 (Jlog q)_r xi_r (phi_plain = sum_r phi_r) and x_r = (s_r xi_r)^2 for E = Z^-1 C, on any scalar type
template <typename S>
__device__ __forceinline__ void cost_phi2(const SE2<S>& Z, const SE2<S>& C, const double* q, const double* s, const Eps2<S>& eps,
                                          S* phi_r, S* x_r, S* a_out, S* xi_out) {
  SE2<S> Zi, E;
  se2_inv(Z, Zi);
  se2_mul(Zi, C, E);
  S xi[3], J[9];
  se2_log_jlog(E, eps, xi, J, true);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const S a = J[3 * r] * S(q[0]) + J[3 * r + 1] * S(q[1]) + J[3 * r + 2] * S(q[2]);
    phi_r[r] = S(0.0) - S(s[r] * s[r]) * a * xi[r];
    x_r[r] = S(s[r] * s[r]) * xi[r] * xi[r];
    if (a_out) { a_out[r] = a; xi_out[r] = xi[r]; }
  }
}

__device__ __forceinline__ void cost_vjp2(const SE2<double>& Z, const SE2<double>& C, const double* q, const double* s,
                                          const Eps2<double>& eps, int loss, double log_radius, double* gZ, double* gs,
                                          double* glr) {
  double phi_r[3], x_r[3], Phi[3], a[3], xi[3];
  cost_phi2<double>(Z, C, q, s, eps, phi_r, x_r, a, xi);
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
  const Eps2<D2> epsd{D2(eps.nz), D2(eps.dnz)};
  const SE2<D2> Cd{D2(C.x), D2(C.y), D2(C.c), D2(C.s)};
  for (int k = 0; k < 4; ++k) {  // one dual evaluation per raw entry [x, y, cos, sin] of Z
    const SE2<D2> Zd{D2(Z.x, k == 0 ? 1.0 : 0.0), D2(Z.y, k == 1 ? 1.0 : 0.0), D2(Z.c, k == 2 ? 1.0 : 0.0),
                     D2(Z.s, k == 3 ? 1.0 : 0.0)};
    D2 phid[3], xd[3];
    cost_phi2<D2>(Zd, Cd, q, s, epsd, phid, xd, nullptr, nullptr);
    double g = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) g += rt.m[r] * phid[r].d + Phi[r] * rt.m_x[r] * xd[r].d;
    gZ[k] = g;
  }
}

template <typename T>
__global__ void __launch_bounds__(64)
pg2_vjp_kernel(thx_pg_structure s, thx_pg_data d, const T* __restrict__ wvec, int64_t ldw, T* __restrict__ g_meas,
               T* __restrict__ g_wb, T* __restrict__ g_tgt, T* __restrict__ g_wp, T* __restrict__ g_lrb, T* __restrict__ g_lrp,
               Eps2<double> eps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int c = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const T* poses = static_cast<const T*>(d.poses);
  const T* wv = wvec + (int64_t)b * ldw;
  double q[3], sw[3], gZ[4], gs[3], glr = 0.0, lr = 0.0;
  int loss = THX_LOSS_NONE;
  SE2<double> Z, C;
  T *outZ, *outS, *outL = nullptr;
  if (c < s.num_edges) {
    const int e = c, i = s.edge_i[e], j = s.edge_j[e];
    const int64_t mB = d.meas_bstride ? B : 1, wB = d.w_between_bstride ? B : 1;
    const SE2<double> Xi = se2_load_d(poses + ((int64_t)i * B + b) * 4), Xj = se2_load_d(poses + ((int64_t)j * B + b) * 4);
    Z = se2_load_d(static_cast<const T*>(d.meas) + ((int64_t)e * mB) * 4 + (int64_t)b * d.meas_bstride);
    const T* wp = static_cast<const T*>(d.w_between) + ((int64_t)e * wB) * 3 + (int64_t)b * d.w_between_bstride;
    SE2<double> Xii, Di;
    se2_inv(Xi, Xii);
    se2_mul(Xii, Xj, C);  // D = v0^-1 v1
    se2_inv(C, Di);
    double Ad[9];
    se2_adjoint(Di, Ad);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double wi0 = (double)wv[3 * i], wi1 = (double)wv[3 * i + 1], wi2 = (double)wv[3 * i + 2];
      q[r] = (double)wv[3 * j + r] - (Ad[3 * r] * wi0 + Ad[3 * r + 1] * wi1 + Ad[3 * r + 2] * wi2);
      sw[r] = (double)wp[r];
    }
    outZ = g_meas + ((int64_t)e * B + b) * 4;
    outS = g_wb + ((int64_t)e * B + b) * 3;
    loss = loss_code(d.robust_between, d.loss_between, e);
    if (loss) lr = load_log_radius<T>(d.log_radius_between, e, b, B, d.log_radius_between_bstride);
    if (d.robust_between) outL = g_lrb ? g_lrb + (int64_t)e * B + b : nullptr;   // (a plain cost of a mixed role: 0)
  } else {
    const int k = c - s.num_edges, p = s.prior_pose[k];
    const int64_t tB = d.prior_target_bstride ? B : 1, wB = d.w_prior_bstride ? B : 1;
    C = se2_load_d(poses + ((int64_t)p * B + b) * 4);
    Z = se2_load_d(static_cast<const T*>(d.prior_target) + ((int64_t)k * tB) * 4 + (int64_t)b * d.prior_target_bstride);
    const T* wp = static_cast<const T*>(d.w_prior) + ((int64_t)k * wB) * 3 + (int64_t)b * d.w_prior_bstride;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      q[r] = (double)wv[3 * p + r];
      sw[r] = (double)wp[r];
    }
    outZ = g_tgt + ((int64_t)k * B + b) * 4;
    outS = g_wp + ((int64_t)k * B + b) * 3;
    loss = loss_code(d.robust_prior, d.loss_prior, k);
    if (loss) lr = load_log_radius<T>(d.log_radius_prior, k, b, B, d.log_radius_prior_bstride);
    if (d.robust_prior) outL = g_lrp ? g_lrp + (int64_t)k * B + b : nullptr;
  }
  cost_vjp2(Z, C, q, sw, eps, loss, lr, gZ, gs, &glr);
  if (outL) *outL = (T)glr;
#pragma unroll
  for (int k = 0; k < 4; ++k) outZ[k] = (T)gZ[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) outS[k] = (T)gs[k];
}

template <typename T>
__global__ void __launch_bounds__(64)
se2_retract_vjp_kernel(const T* __restrict__ poses, const T* __restrict__ delta, int64_t ldd, T step,
                       const T* __restrict__ gout, T* __restrict__ gdelta, int64_t ldg, int P, int B, Eps2<double> eps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B) return;
  const SE2<double> X = se2_load_d(poses + ((int64_t)p * B + b) * 4);
  const T* go = gout + ((int64_t)p * B + b) * 4;
  const double G[4] = {(double)go[0], (double)go[1], (double)go[2], (double)go[3]};
  double xi[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) xi[i] = (double)(delta[(int64_t)b * ldd + 3 * p + i] * step);
  const Eps2<D2> epsd{D2(eps.nz), D2(eps.dnz)};
  const SE2<D2> Xd{D2(X.x), D2(X.y), D2(X.c), D2(X.s)};
  for (int k = 0; k < 3; ++k) {
    const D2 xid[3] = {D2(xi[0], k == 0 ? 1.0 : 0.0), D2(xi[1], k == 1 ? 1.0 : 0.0), D2(xi[2], k == 2 ? 1.0 : 0.0)};
    SE2<D2> Ex, Y;
    se2_exp<D2>(xid, epsd, Ex, nullptr);
    se2_mul(Xd, Ex, Y);
    gdelta[(int64_t)b * ldg + 3 * p + k] = (T)((G[0] * Y.x.d + G[1] * Y.y.d + G[2] * Y.c.d + G[3] * Y.s.d) * (double)step);
  }
}

static inline Eps2<double> eps2d(const thx_se2_eps* e, int dtype) {
  return dtype == THX_F32 ? Eps2<double>{(double)(float)e->near_zero, (double)(float)e->d_near_zero}
                          : Eps2<double>{e->near_zero, e->d_near_zero};
}

}  // namespace thx

using namespace thx;

extern "C" {

int thx_pg2_vjp(const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw, void* grad_meas,
                void* grad_w_between, void* grad_prior_target, void* grad_w_prior, void* grad_log_radius_between,
                void* grad_log_radius_prior, int dtype, const thx_se2_eps* eps, void* stream) {
  if (!s || !d || !w || !eps) return fail("thx_pg2_vjp: null argument");
  if (s->num_edges > 0 && (!grad_meas || !grad_w_between)) return fail("thx_pg2_vjp: null edge gradient buffer");
  if (s->num_priors > 0 && (!grad_prior_target || !grad_w_prior)) return fail("thx_pg2_vjp: null prior gradient buffer");
  if (ldw < 3 * (int64_t)s->num_poses) return fail("thx_pg2_vjp: ldw < n");
  if (const char* why = check_robust(d)) return fail(why);
  dim3 grid((d->batch + 63) / 64, s->num_edges + s->num_priors), block(64);
  if (grid.y == 0) return 0;
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(pg2_vjp_kernel<float>, grid, block, 0, as_stream(stream), *s, *d, (const float*)w, ldw,
                                  (float*)grad_meas, (float*)grad_w_between, (float*)grad_prior_target, (float*)grad_w_prior,
                                  (float*)grad_log_radius_between, (float*)grad_log_radius_prior, eps2d(eps, dtype)),
               hipLaunchKernelGGL(pg2_vjp_kernel<double>, grid, block, 0, as_stream(stream), *s, *d, (const double*)w, ldw,
                                  (double*)grad_meas, (double*)grad_w_between, (double*)grad_prior_target,
                                  (double*)grad_w_prior, (double*)grad_log_radius_between, (double*)grad_log_radius_prior,
                                  eps2d(eps, dtype)));
  return check_launch("thx_pg2_vjp");
}

int thx_se2_retract_vjp(const void* poses, const void* delta, int64_t ldd, double step, const void* grad_out,
                        void* grad_delta, int64_t ldg, int32_t P, int32_t B, int dtype, const thx_se2_eps* eps, void* stream) {
  if (!poses || !delta || !grad_out || !grad_delta || !eps || P <= 0 || B <= 0) return fail("bad se2_retract_vjp args");
  dim3 grid((B + 63) / 64, P), block(64);
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(se2_retract_vjp_kernel<float>, grid, block, 0, as_stream(stream), (const float*)poses,
                                  (const float*)delta, ldd, (float)step, (const float*)grad_out, (float*)grad_delta, ldg, P,
                                  B, eps2d(eps, dtype)),
               hipLaunchKernelGGL(se2_retract_vjp_kernel<double>, grid, block, 0, as_stream(stream), (const double*)poses,
                                  (const double*)delta, ldd, step, (const double*)grad_out, (double*)grad_delta, ldg, P, B,
                                  eps2d(eps, dtype)));
  return check_launch("thx_se2_retract_vjp");
}

}  // extern "C"
