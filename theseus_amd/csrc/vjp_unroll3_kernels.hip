// BackwardMode.UNROLL / TRUNCATED on SE2 / SO3 pose graphs: the 3-dof twins of vjp_unroll_kernels.hip over unroll_g3.cuh.
// One lane per (cost, problem); double arithmetic whatever the storage type; per-cost outputs (the host sums the pose gradients).
#include "common.cuh"
#include "unroll_g3.cuh"

namespace thx {

template <typename T, typename G>
__global__ void __launch_bounds__(64)
pg3_unroll_vjp_kernel(thx_pg_structure s, thx_pg_data d, const T* __restrict__ wvec, int64_t ldw, const T* __restrict__ dvec,
                      int64_t ldd, const T* __restrict__ ell_damping, T* __restrict__ g_pose_i, T* __restrict__ g_pose_j,
                      T* __restrict__ g_meas, T* __restrict__ g_wb, T* __restrict__ g_pose_p, T* __restrict__ g_tgt,
                      T* __restrict__ g_wp, T* __restrict__ g_lrb, T* __restrict__ g_lrp, typename G::EpsD eps) {
  constexpr int NR = G::NR;
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int c = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const T* poses = static_cast<const T*>(d.poses);
  const T* wv = wvec + (int64_t)b * ldw;
  const T* dv = dvec + (int64_t)b * ldd;
  const double lam = ell_damping ? (double)ell_damping[b] : 0.0;
  double ri[NR], rj[NR], rz[NR], sw[3], wi[3] = {0, 0, 0}, wj[3], di[3] = {0, 0, 0}, dj[3], g[3 * NR], gs[3], glr = 0.0;
  auto load = [&](const T* p, double* r) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NR; ++k) r[k] = (double)p[k];
  };
  if (c < s.num_edges) {
    const int e = c, i = s.edge_i[e], j = s.edge_j[e];
    const int64_t mB = d.meas_bstride ? B : 1, wB = d.w_between_bstride ? B : 1;
    load(poses + ((int64_t)i * B + b) * NR, ri);
    load(poses + ((int64_t)j * B + b) * NR, rj);
    load(static_cast<const T*>(d.meas) + ((int64_t)e * mB) * NR + (int64_t)b * d.meas_bstride, rz);
    const T* wp = static_cast<const T*>(d.w_between) + ((int64_t)e * wB) * 3 + (int64_t)b * d.w_between_bstride;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      wi[k] = (double)wv[3 * i + k];
      wj[k] = (double)wv[3 * j + k];
      di[k] = (double)dv[3 * i + k];
      dj[k] = (double)dv[3 * j + k];
      sw[k] = (double)wp[k];
    }
    const int loss = loss_code(d.robust_between, d.loss_between, e);
    const double lr = loss ? load_log_radius<T>(d.log_radius_between, e, b, B, d.log_radius_between_bstride) : 0.0;
    unroll3_vjp<G, true>(ri, rj, rz, sw, wi, wj, di, dj, eps, lam, loss, lr, g, gs, &glr);
    T* oi = g_pose_i + ((int64_t)e * B + b) * NR;
    T* oj = g_pose_j + ((int64_t)e * B + b) * NR;
    T* oz = g_meas + ((int64_t)e * B + b) * NR;
    T* os = g_wb + ((int64_t)e * B + b) * 3;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      oi[k] = (T)g[k];
      oj[k] = (T)g[NR + k];
      oz[k] = (T)g[2 * NR + k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) os[k] = (T)gs[k];
    if (d.robust_between && g_lrb) g_lrb[(int64_t)e * B + b] = (T)glr;
  } else {
    const int k0 = c - s.num_edges, p = s.prior_pose[k0];
    const int64_t tB = d.prior_target_bstride ? B : 1, wB = d.w_prior_bstride ? B : 1;
    load(poses + ((int64_t)p * B + b) * NR, rj);
    load(static_cast<const T*>(d.prior_target) + ((int64_t)k0 * tB) * NR + (int64_t)b * d.prior_target_bstride, rz);
    const T* wp = static_cast<const T*>(d.w_prior) + ((int64_t)k0 * wB) * 3 + (int64_t)b * d.w_prior_bstride;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      wj[r] = (double)wv[3 * p + r];
      dj[r] = (double)dv[3 * p + r];
      sw[r] = (double)wp[r];
    }
    const int loss = loss_code(d.robust_prior, d.loss_prior, k0);
    const double lr = loss ? load_log_radius<T>(d.log_radius_prior, k0, b, B, d.log_radius_prior_bstride) : 0.0;
    unroll3_vjp<G, false>(rj, rj, rz, sw, wi, wj, di, dj, eps, lam, loss, lr, g, gs, &glr);
    T* ox = g_pose_p + ((int64_t)k0 * B + b) * NR;
    T* ot = g_tgt + ((int64_t)k0 * B + b) * NR;
    T* os = g_wp + ((int64_t)k0 * B + b) * 3;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      ox[k] = (T)g[NR + k];
      ot[k] = (T)g[2 * NR + k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) os[k] = (T)gs[k];
    if (d.robust_prior && g_lrp) g_lrp[(int64_t)k0 * B + b] = (T)glr;
  }
}

template <typename G>
static int launch_unroll3(const char* what, const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw,
                          const void* delta, int64_t ldd, const void* ell, void* gpi, void* gpj, void* gm, void* gwb, void* gpp,
                          void* gt, void* gwp, void* glb, void* glp, int dtype, typename G::EpsD eps, void* stream) {
  if (!s || !d || !w || !delta) return fail("thx_pg*_unroll_vjp: null argument");
  if (s->num_edges > 0 && (!gpi || !gpj || !gm || !gwb)) return fail("thx_pg*_unroll_vjp: null edge gradient buffer");
  if (s->num_priors > 0 && (!gpp || !gt || !gwp)) return fail("thx_pg*_unroll_vjp: null prior gradient buffer");
  if (ldw < 3 * (int64_t)s->num_poses || ldd < 3 * (int64_t)s->num_poses) return fail("thx_pg*_unroll_vjp: ldw / ldd < n");
  if (const char* why = check_robust(d)) return fail(why);
  dim3 grid((d->batch + 63) / 64, s->num_edges + s->num_priors), block(64);
  if (grid.y == 0) return 0;
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL((pg3_unroll_vjp_kernel<float, G>), grid, block, 0, as_stream(stream), *s, *d, (const float*)w, ldw,
                                  (const float*)delta, ldd, (const float*)ell, (float*)gpi, (float*)gpj, (float*)gm, (float*)gwb,
                                  (float*)gpp, (float*)gt, (float*)gwp, (float*)glb, (float*)glp, eps),
               hipLaunchKernelGGL((pg3_unroll_vjp_kernel<double, G>), grid, block, 0, as_stream(stream), *s, *d, (const double*)w,
                                  ldw, (const double*)delta, ldd, (const double*)ell, (double*)gpi, (double*)gpj, (double*)gm,
                                  (double*)gwb, (double*)gpp, (double*)gt, (double*)gwp, (double*)glb, (double*)glp, eps));
  return check_launch(what);
}

}  // namespace thx

using namespace thx;

extern "C" {

int thx_pg2_unroll_vjp(const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw, const void* delta, int64_t ldd,
                       const void* ellipsoidal_damping, void* grad_pose_i, void* grad_pose_j, void* grad_meas, void* grad_w_between,
                       void* grad_pose_prior, void* grad_prior_target, void* grad_w_prior, void* grad_log_radius_between,
                       void* grad_log_radius_prior, int dtype, const thx_se2_eps* eps, void* stream) {
  if (!eps) return fail("thx_pg2_unroll_vjp: null eps");
  return launch_unroll3<UG_SE2>("thx_pg2_unroll_vjp", s, d, w, ldw, delta, ldd, ellipsoidal_damping, grad_pose_i, grad_pose_j,
                                grad_meas, grad_w_between, grad_pose_prior, grad_prior_target, grad_w_prior, grad_log_radius_between,
                                grad_log_radius_prior, dtype, Eps2<double>{eps->near_zero, eps->d_near_zero}, stream);
}

int thx_pgso3_unroll_vjp(const thx_pg_structure* s, const thx_pg_data* d, const void* w, int64_t ldw, const void* delta, int64_t ldd,
                         const void* ellipsoidal_damping, void* grad_pose_i, void* grad_pose_j, void* grad_meas,
                         void* grad_w_between, void* grad_pose_prior, void* grad_prior_target, void* grad_w_prior,
                         void* grad_log_radius_between, void* grad_log_radius_prior, int dtype, const thx_lie_eps* eps, void* stream) {
  if (!eps) return fail("thx_pgso3_unroll_vjp: null eps");
  return launch_unroll3<UG_SO3>("thx_pgso3_unroll_vjp", s, d, w, ldw, delta, ldd, ellipsoidal_damping, grad_pose_i, grad_pose_j,
                                grad_meas, grad_w_between, grad_pose_prior, grad_prior_target, grad_w_prior, grad_log_radius_between,
                                grad_log_radius_prior, dtype, Eps<double>{eps->near_zero, eps->d_near_zero, eps->near_pi}, stream);
}

}  // extern "C"
