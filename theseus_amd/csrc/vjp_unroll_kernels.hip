// BackwardMode.UNROLL / TRUNCATED on SE3 pose graphs: the per-cost VJP of one differentiated iteration (unroll_se3.cuh).
// One lane per (cost, problem), batch index fastest across the wave; double arithmetic whatever the storage type.  Outputs are
// PER COST (the host sums the pose gradients of a pose's incident costs in a fixed order: no atomics, bit-reproducible).
#include "common.cuh"
#include "unroll_se3.cuh"
#include "vjp_se3.cuh"   // load_se3_any

namespace thx {

template <typename T>
__global__ void __launch_bounds__(64)
pg_unroll_vjp_kernel(thx_pg_structure s, thx_pg_data d, const T* __restrict__ wvec, int64_t ldw, const T* __restrict__ dvec,
                     int64_t ldd, const T* __restrict__ ell_damping, T* __restrict__ g_pose_i, T* __restrict__ g_pose_j, T* __restrict__ g_meas,
                     T* __restrict__ g_wb, T* __restrict__ g_pose_p, T* __restrict__ g_tgt, T* __restrict__ g_wp,
                     T* __restrict__ g_lrb, T* __restrict__ g_lrp, Eps<T> eps_t) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int c = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const Eps<double> eps{(double)eps_t.nz, (double)eps_t.dnz, (double)eps_t.npi};
  const T* poses = static_cast<const T*>(d.poses);
  const T* wv = wvec + (int64_t)b * ldw;
  const T* dv = dvec + (int64_t)b * ldd;
  double sw[6], gs[6];
  const double lam = ell_damping ? (double)ell_damping[b] : 0.0;   // ellipsoidal damping: lambda_b (null: spherical / none)
  if (c < s.num_edges) {
    const int e = c, i = s.edge_i[e], j = s.edge_j[e];
    const int64_t mB = d.meas_bstride ? B : 1, wB = d.w_between_bstride ? B : 1;
    SE3<double> Xi, Xj, Z;
    load_se3_any(poses + ((int64_t)i * B + b) * 12, Xi);
    load_se3_any(poses + ((int64_t)j * B + b) * 12, Xj);
    load_se3_any(static_cast<const T*>(d.meas) + ((int64_t)e * mB) * 12 + (int64_t)b * d.meas_bstride, Z);
    const T* wp = static_cast<const T*>(d.w_between) + ((int64_t)e * wB) * 6 + (int64_t)b * d.w_between_bstride;
    double wi[6], wj[6], di[6], dj[6], gXi[12], gXj[12], gZ[12];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      wi[k] = (double)wv[6 * i + k];
      wj[k] = (double)wv[6 * j + k];
      di[k] = (double)dv[6 * i + k];
      dj[k] = (double)dv[6 * j + k];
      sw[k] = (double)wp[k];
    }
    const int loss = loss_code(d.robust_between, d.loss_between, e);
    const double lr = loss ? load_log_radius<T>(d.log_radius_between, e, b, B, d.log_radius_between_bstride) : 0.0;
    double glr = 0.0;
    unroll_edge_vjp(Xi, Xj, Z, sw, wi, wj, di, dj, eps, gXi, gXj, gZ, gs, lam, loss, lr, &glr);
    if (d.robust_between && g_lrb) g_lrb[(int64_t)e * B + b] = (T)glr;   // (a plain cost of a mixed role: 0)
    T* oi = g_pose_i + ((int64_t)e * B + b) * 12;
    T* oj = g_pose_j + ((int64_t)e * B + b) * 12;
    T* oz = g_meas + ((int64_t)e * B + b) * 12;
    T* os = g_wb + ((int64_t)e * B + b) * 6;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      oi[k] = (T)gXi[k];
      oj[k] = (T)gXj[k];
      oz[k] = (T)gZ[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) os[k] = (T)gs[k];
  } else {
    const int k0 = c - s.num_edges, p = s.prior_pose[k0];
    const int64_t tB = d.prior_target_bstride ? B : 1, wB = d.w_prior_bstride ? B : 1;
    SE3<double> X, Tg;
    load_se3_any(poses + ((int64_t)p * B + b) * 12, X);
    load_se3_any(static_cast<const T*>(d.prior_target) + ((int64_t)k0 * tB) * 12 + (int64_t)b * d.prior_target_bstride, Tg);
    const T* wp = static_cast<const T*>(d.w_prior) + ((int64_t)k0 * wB) * 6 + (int64_t)b * d.w_prior_bstride;
    double w6[6], d6[6], gX[12], gT[12];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      w6[r] = (double)wv[6 * p + r];
      d6[r] = (double)dv[6 * p + r];
      sw[r] = (double)wp[r];
    }
    const int loss = loss_code(d.robust_prior, d.loss_prior, k0);
    const double lr = loss ? load_log_radius<T>(d.log_radius_prior, k0, b, B, d.log_radius_prior_bstride) : 0.0;
    double glr = 0.0;
    unroll_prior_vjp(X, Tg, sw, w6, d6, eps, gX, gT, gs, lam, loss, lr, &glr);
    if (d.robust_prior && g_lrp) g_lrp[(int64_t)k0 * B + b] = (T)glr;
    T* ox = g_pose_p + ((int64_t)k0 * B + b) * 12;
    T* ot = g_tgt + ((int64_t)k0 * B + b) * 12;
    T* os = g_wp + ((int64_t)k0 * B + b) * 6;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      ox[k] = (T)gX[k];
      ot[k] = (T)gT[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) os[k] = (T)gs[k];
  }
}

}  // namespace thx

using namespace thx;

extern "C" {

int thx_pg_unroll_vjp(const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw, const void* delta, int64_t ldd,
                      const void* ellipsoidal_damping, void* grad_pose_i, void* grad_pose_j, void* grad_meas, void* grad_w_between, void* grad_pose_prior,
                      void* grad_prior_target, void* grad_w_prior, void* grad_log_radius_between, void* grad_log_radius_prior, int dtype,
                      const thx_lie_eps* eps, void* stream) {
  if (!s || !d || !w || !delta || !eps) return fail("thx_pg_unroll_vjp: null argument");
  if (s->num_edges > 0 && (!grad_pose_i || !grad_pose_j || !grad_meas || !grad_w_between))
    return fail("thx_pg_unroll_vjp: null edge gradient buffer");
  if (s->num_priors > 0 && (!grad_pose_prior || !grad_prior_target || !grad_w_prior))
    return fail("thx_pg_unroll_vjp: null prior gradient buffer");
  if (ldw < 6 * (int64_t)s->num_poses || ldd < 6 * (int64_t)s->num_poses) return fail("thx_pg_unroll_vjp: ldw / ldd < n");
  if (const char* why = check_robust(d)) return fail(why);
  dim3 grid((d->batch + 63) / 64, s->num_edges + s->num_priors), block(64);
  if (grid.y == 0) return 0;
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(pg_unroll_vjp_kernel<float>, grid, block, 0, as_stream(stream), *s, *d, (const float*)w, ldw,
                                  (const float*)delta, ldd, (const float*)ellipsoidal_damping, (float*)grad_pose_i, (float*)grad_pose_j, (float*)grad_meas,
                                  (float*)grad_w_between, (float*)grad_pose_prior, (float*)grad_prior_target, (float*)grad_w_prior,
                                  (float*)grad_log_radius_between, (float*)grad_log_radius_prior, make_eps<float>(eps)),
               hipLaunchKernelGGL(pg_unroll_vjp_kernel<double>, grid, block, 0, as_stream(stream), *s, *d, (const double*)w, ldw,
                                  (const double*)delta, ldd, (const double*)ellipsoidal_damping, (double*)grad_pose_i, (double*)grad_pose_j, (double*)grad_meas,
                                  (double*)grad_w_between, (double*)grad_pose_prior, (double*)grad_prior_target,
                                  (double*)grad_w_prior, (double*)grad_log_radius_between, (double*)grad_log_radius_prior,
                                  make_eps<double>(eps)));
  return check_launch("thx_pg_unroll_vjp");
}

}  // extern "C"
