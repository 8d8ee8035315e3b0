// BackwardMode.UNROLL / TRUNCATED on bundle-adjustment objectives: the per-cost VJP of one differentiated iteration
// (unroll_ba.cuh).  One lane per (cost, problem), batch index fastest across the wave; double arithmetic whatever the storage
// type.  Outputs are PER COST: the host sums the camera / point gradients over each variable's costs.
#include "common.cuh"
#include "unroll_ba.cuh"
#include "vjp_se3.cuh"   // load_se3_any

namespace thx {

template <typename T>
__global__ void __launch_bounds__(64)
ba_unroll_vjp_kernel(thx_ba_structure s, thx_ba_data d, const T* __restrict__ wvec, int64_t ldw, const T* __restrict__ dvec,
                     int64_t ldd, const T* __restrict__ ell_damping, thx_ba_unroll_grads out, Eps<double> eps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int c = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const T* wv = wvec + (int64_t)b * ldw;
  const T* dv = dvec + (int64_t)b * ldd;
  const int nc = 6 * s.num_cams;
  const double lam = ell_damping ? (double)ell_damping[b] : 0.0;
  if (c < s.num_obs) {
    const int o = c, cam_i = s.obs_cam[o], pt_i = s.obs_pt[o];
    SE3<double> cam;
    load_se3_any(static_cast<const T*>(d.cams) + ((int64_t)cam_i * B + b) * 12, cam);
    const T* Xp = static_cast<const T*>(d.points) + ((int64_t)pt_i * B + b) * 3;
    const double X[3] = {(double)Xp[0], (double)Xp[1], (double)Xp[2]};
    const T* fp = static_cast<const T*>(d.feat) + ((int64_t)o * (d.feat_bstride ? B : 1)) * 2 + (int64_t)b * d.feat_bstride;
    const T* wp = static_cast<const T*>(d.w_obs) + ((int64_t)o * (d.w_obs_bstride ? B : 1)) * 2 + (int64_t)b * d.w_obs_bstride;
    const int64_t ci = (int64_t)cam_i * (d.calib_bstride ? B : 1) + (int64_t)b * d.calib_bstride;
    const double feat[2] = {(double)fp[0], (double)fp[1]}, sw[2] = {(double)wp[0], (double)wp[1]};
    const double f = (double)static_cast<const T*>(d.focal)[ci], k1 = (double)static_cast<const T*>(d.k1)[ci],
                 k2 = (double)static_cast<const T*>(d.k2)[ci];
    double wc[6], wq[3], dc[6], dq[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      wc[k] = (double)wv[6 * cam_i + k];
      dc[k] = (double)dv[6 * cam_i + k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      wq[k] = (double)wv[nc + 3 * pt_i + k];
      dq[k] = (double)dv[nc + 3 * pt_i + k];
    }
    const double lr = d.robust_obs ? load_log_radius<T>(d.log_radius_obs, o, b, B, d.log_radius_obs_bstride) : 0.0;
    double gcam[12], gX[3], gfeat[2], gs[2], gcal[3], glr;
    unroll_reproj_vjp(cam, X, feat, f, k1, k2, sw, wc, wq, dc, dq, lam, d.robust_obs, lr, gcam, gX, gfeat, gs, gcal, &glr);
    const int64_t ob = (int64_t)o * B + b;
#pragma unroll
    for (int k = 0; k < 12; ++k) static_cast<T*>(out.cam_obs)[ob * 12 + k] = (T)gcam[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) static_cast<T*>(out.pt_obs)[ob * 3 + k] = (T)gX[k];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      static_cast<T*>(out.feat)[ob * 2 + r] = (T)gfeat[r];
      static_cast<T*>(out.w_obs)[ob * 2 + r] = (T)gs[r];
    }
    static_cast<T*>(out.focal)[ob] = (T)gcal[0];
    static_cast<T*>(out.k1)[ob] = (T)gcal[1];
    static_cast<T*>(out.k2)[ob] = (T)gcal[2];
    if (out.log_radius_obs) static_cast<T*>(out.log_radius_obs)[ob] = (T)glr;
  } else if (c < s.num_obs + s.num_cam_priors) {
    const int k = c - s.num_obs, cam_i = s.cam_prior_cam[k];
    SE3<double> Z, C;
    load_se3_any(static_cast<const T*>(d.cams) + ((int64_t)cam_i * B + b) * 12, C);
    load_se3_any(static_cast<const T*>(d.cam_prior_target) + ((int64_t)k * (d.cam_prior_target_bstride ? B : 1)) * 12 +
                     (int64_t)b * d.cam_prior_target_bstride, Z);
    const T* wp = static_cast<const T*>(d.w_cam_prior) + ((int64_t)k * (d.w_cam_prior_bstride ? B : 1)) * 6 +
                  (int64_t)b * d.w_cam_prior_bstride;
    double w6[6], d6[6], sw[6], gC[12], gZ[12], gs[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      w6[r] = (double)wv[6 * cam_i + r];
      d6[r] = (double)dv[6 * cam_i + r];
      sw[r] = (double)wp[r];
    }
    unroll_prior_vjp(C, Z, sw, w6, d6, eps, gC, gZ, gs, lam);
    const int64_t kb = (int64_t)k * B + b;
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      static_cast<T*>(out.cam_prior_cam)[kb * 12 + r] = (T)gC[r];
      static_cast<T*>(out.cam_prior_target)[kb * 12 + r] = (T)gZ[r];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) static_cast<T*>(out.w_cam_prior)[kb * 6 + r] = (T)gs[r];
  } else {
    const int k = c - s.num_obs - s.num_cam_priors, pt_i = s.pt_prior_pt[k];
    const T* Xp = static_cast<const T*>(d.points) + ((int64_t)pt_i * B + b) * 3;
    const T* tp = static_cast<const T*>(d.pt_prior_target) + ((int64_t)k * (d.pt_prior_target_bstride ? B : 1)) * 3 +
                  (int64_t)b * d.pt_prior_target_bstride;
    const T* wp = static_cast<const T*>(d.w_pt_prior) + ((int64_t)k * (d.w_pt_prior_bstride ? B : 1)) * 3 +
                  (int64_t)b * d.w_pt_prior_bstride;
    double X[3], t3[3], sw[3], w3[3], d3[3], gX[3], gT[3], gs[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      X[r] = (double)Xp[r];
      t3[r] = (double)tp[r];
      sw[r] = (double)wp[r];
      w3[r] = (double)wv[nc + 3 * pt_i + r];
      d3[r] = (double)dv[nc + 3 * pt_i + r];
    }
    unroll_pt_prior_vjp(X, t3, sw, w3, d3, lam, gX, gT, gs);
    const int64_t kb = (int64_t)k * B + b;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      static_cast<T*>(out.pt_prior_pt)[kb * 3 + r] = (T)gX[r];
      static_cast<T*>(out.pt_prior_target)[kb * 3 + r] = (T)gT[r];
      static_cast<T*>(out.w_pt_prior)[kb * 3 + r] = (T)gs[r];
    }
  }
}

}  // namespace thx

using namespace thx;

extern "C" {

int thx_ba_unroll_vjp(const thx_ba_structure* s, const thx_ba_data* d, const void* w, int64_t ldw, const void* delta, int64_t ldd,
                      const void* ellipsoidal_damping, const thx_ba_unroll_grads* out, int dtype, const thx_lie_eps* eps,
                      void* stream) {
  if (!s || !d || !w || !delta || !out || !eps) return fail("thx_ba_unroll_vjp: null argument");
  const int64_t n = 6 * (int64_t)s->num_cams + 3 * (int64_t)s->num_points;
  if (ldw < n || ldd < n) return fail("thx_ba_unroll_vjp: ldw / ldd < n");
  if (s->num_obs > 0 && (!out->cam_obs || !out->pt_obs || !out->feat || !out->w_obs || !out->focal || !out->k1 || !out->k2))
    return fail("thx_ba_unroll_vjp: null observation gradient buffer");
  if (s->num_cam_priors > 0 && (!out->cam_prior_cam || !out->cam_prior_target || !out->w_cam_prior))
    return fail("thx_ba_unroll_vjp: null camera-prior gradient buffer");
  if (s->num_pt_priors > 0 && (!out->pt_prior_pt || !out->pt_prior_target || !out->w_pt_prior))
    return fail("thx_ba_unroll_vjp: null point-prior gradient buffer");
  if (d->robust_obs && !d->log_radius_obs) return fail("thx_ba_unroll_vjp: robust cost without log_loss_radius");
  if (!loss_code_valid(d->robust_obs)) return fail("thx_ba_unroll_vjp: bad loss kind");
  dim3 grid((d->batch + 63) / 64, s->num_obs + s->num_cam_priors + s->num_pt_priors), block(64);
  if (grid.y == 0) return 0;
  const Eps<double> e = dtype == THX_F32 ? Eps<double>{(double)(float)eps->near_zero, (double)(float)eps->d_near_zero, (double)(float)eps->near_pi}
                                         : Eps<double>{eps->near_zero, eps->d_near_zero, eps->near_pi};
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(ba_unroll_vjp_kernel<float>, grid, block, 0, as_stream(stream), *s, *d, (const float*)w, ldw,
                                  (const float*)delta, ldd, (const float*)ellipsoidal_damping, *out, e),
               hipLaunchKernelGGL(ba_unroll_vjp_kernel<double>, grid, block, 0, as_stream(stream), *s, *d, (const double*)w, ldw,
                                  (const double*)delta, ldd, (const double*)ellipsoidal_damping, *out, e));
  return check_launch("thx_ba_unroll_vjp");
}

}  // extern "C"
