// Implicit backward for bundle-adjustment objectives (BackwardMode.IMPLICIT, theseus/optimizer/nonlinear/
// nonlinear_least_squares.py:121-135,265-292; examples/bundle_adjustment.py:184-215 learns log_loss_radius through it):
//   X_new = retract(X, delta),  delta = H^-1 g(theta)  with H detached;  backward: grad_X_new -> grad_delta (thx_se3_retract_vjp on
//   the camera columns, identity on the point columns), w = H^-1 grad_delta with the cached Schur factor, and THIS kernel:
//   grad_theta of phi = w^T g, one lane per (cost, problem).
// Reprojection cost o (embodied/measurements/reprojection.py:54-94), camera c, point p, RAW residual eps_r and Jacobian J_r,
// row weights s_r, robust rescale^2 m(x, log_radius), x = sum_r (s_r eps_r)^2 (core/robust_cost_function.py:115-135):
//   phi_o = - m sum_r s_r^2 alpha_r eps_r ,   alpha = Jc w_c + Jp w_p
//   d/d feature u_r   : eps_r = pi_r - u_r, J independent of u  ->  m s_r^2 alpha_r - phi_plain m_x 2 s_r^2 eps_r
//   d/d s_r           : - m 2 s_r alpha_r eps_r + phi_plain m_x 2 s_r eps_r^2
//   d/d log_radius    : phi_plain m_l
//   d/d focal, k1, k2 : dual numbers through the same closed forms (residual AND Jacobian depend on them)
// SE3 Difference priors on cameras: cost_vjp of vjp_se3.cuh (target, weights); Point3 Difference priors: closed form.
#include "vjp_se3.cuh"

namespace thx {

// raw (unweighted) reprojection residual eps (2) and alpha = Jc qc + Jp qp (2) on scalar type S (double | Dual<double>) for
// the calibration; camera, point, feature and q are plain doubles
template <typename S>
__device__ __forceinline__ void reproj_alpha_eps(const SE3<double>& cam, const double* X, const double* feat, S f, S k1, S k2,
                                                 const double* qc, const double* qp, S* alpha, S* eps) {
  double pc[3];
  mat3_vec(cam.R, X, pc);
#pragma unroll
  for (int i = 0; i < 3; ++i) pc[i] += cam.t[i];
  const double iz = 1.0 / pc[2];
  const double proj[2] = {-pc[0] * iz, -pc[1] * iz};
  const double q = proj[0] * proj[0] + proj[1] * proj[1];
  const S factor = f * (S(1.0) + S(q) * (k1 + S(q) * k2));
  const S dfactor = f * (k1 + S(2.0 * q) * k2);
  eps[0] = S(proj[0]) * factor - S(feat[0]);
  eps[1] = S(proj[1]) * factor - S(feat[1]);
  // d pc / d [xi_cam (6) | X (3)] . [qc | qp]  =  R qc_lin - R hat(X) qc_ang + R qp      (se3_impl.py:757-777)
  double hq[3], v[3], dpc[3];
  cross3(X, qc + 3, hq);                       // hat(X) qc_ang
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = qc[i] - hq[i] + qp[i];
  mat3_vec(cam.R, v, dpc);
  const double j2 = dpc[2] * iz;
  const double pj0 = (pc[0] * j2 - dpc[0]) * iz, pj1 = (pc[1] * j2 - dpc[1]) * iz;   // d proj . q
  const double qj = 2.0 * (proj[0] * pj0 + proj[1] * pj1);
  alpha[0] = S(pj0) * factor + S(proj[0] * qj) * dfactor;
  alpha[1] = S(pj1) * factor + S(proj[1] * qj) * dfactor;
}

template <typename T>
__global__ void __launch_bounds__(64)
ba_vjp_kernel(thx_ba_structure s, thx_ba_data d, const T* __restrict__ wvec, int64_t ldw, T* __restrict__ g_feat,
              T* __restrict__ g_wobs, T* __restrict__ g_focal, T* __restrict__ g_k1, T* __restrict__ g_k2, T* __restrict__ g_lr,
              T* __restrict__ g_cpt, T* __restrict__ g_wcp, T* __restrict__ g_ppt, T* __restrict__ g_wpp, Eps<double> eps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int c = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const T* wv = wvec + (int64_t)b * ldw;
  const int nc = 6 * s.num_cams;
  using D2 = Dual<double>;
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
    double qc[6], qp[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) qc[k] = (double)wv[6 * cam_i + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) qp[k] = (double)wv[nc + 3 * pt_i + k];
    double al[2], ep[2];
    reproj_alpha_eps<double>(cam, X, feat, f, k1, k2, qc, qp, al, ep);
    double phi_r[2], x_r[2], Phi[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      phi_r[r] = -sw[r] * sw[r] * al[r] * ep[r];
      x_r[r] = sw[r] * sw[r] * ep[r] * ep[r];
    }
    RobustTerms<2> rt;   // robust.cuh
    rt.eval(d.robust_obs, x_r, d.robust_obs ? load_log_radius<T>(d.log_radius_obs, o, b, B, d.log_radius_obs_bstride) : 0.0);
    rt.group(phi_r, Phi);
    const int64_t ob = (int64_t)o * B + b;
    if (g_lr) g_lr[ob] = (T)(phi_r[0] * rt.m_l[0] + phi_r[1] * rt.m_l[1]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (g_feat) g_feat[ob * 2 + r] = (T)(rt.m[r] * sw[r] * sw[r] * al[r] - Phi[r] * rt.m_x[r] * 2.0 * sw[r] * sw[r] * ep[r]);
      if (g_wobs) g_wobs[ob * 2 + r] = (T)(-rt.m[r] * 2.0 * sw[r] * al[r] * ep[r] + Phi[r] * rt.m_x[r] * 2.0 * sw[r] * ep[r] * ep[r]);
    }
    if (g_focal) {  // one dual evaluation per calibration parameter
      T* outs[3] = {g_focal, g_k1, g_k2};
      for (int k = 0; k < 3; ++k) {
        D2 ald[2], epd[2];
        reproj_alpha_eps<D2>(cam, X, feat, D2(f, k == 0 ? 1.0 : 0.0), D2(k1, k == 1 ? 1.0 : 0.0), D2(k2, k == 2 ? 1.0 : 0.0), qc, qp,
                             ald, epd);
        double g = 0.0;
#pragma unroll
        for (int r = 0; r < 2; ++r)
          g += rt.m[r] * (-sw[r] * sw[r] * (ald[r].d * ep[r] + al[r] * epd[r].d)) + Phi[r] * rt.m_x[r] * (2.0 * sw[r] * sw[r] * ep[r] * epd[r].d);
        outs[k][ob] = (T)g;
      }
    }
  } else if (c < s.num_obs + s.num_cam_priors) {
    const int k = c - s.num_obs, cam_i = s.cam_prior_cam[k];
    SE3<double> Z, C;
    load_se3_any(static_cast<const T*>(d.cams) + ((int64_t)cam_i * B + b) * 12, C);
    load_se3_any(static_cast<const T*>(d.cam_prior_target) + ((int64_t)k * (d.cam_prior_target_bstride ? B : 1)) * 12 +
                     (int64_t)b * d.cam_prior_target_bstride, Z);
    const T* wp = static_cast<const T*>(d.w_cam_prior) + ((int64_t)k * (d.w_cam_prior_bstride ? B : 1)) * 6 +
                  (int64_t)b * d.w_cam_prior_bstride;
    double q[6], sw[6], gZ[12], gs[6], glr;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      q[r] = (double)wv[6 * cam_i + r];
      sw[r] = (double)wp[r];
    }
    cost_vjp(Z, C, q, sw, eps, THX_LOSS_NONE, 0.0, gZ, gs, &glr);
    const int64_t kb = (int64_t)k * B + b;
    if (g_cpt) {
#pragma unroll
      for (int r = 0; r < 12; ++r) g_cpt[kb * 12 + r] = (T)gZ[r];
    }
    if (g_wcp) {
#pragma unroll
      for (int r = 0; r < 6; ++r) g_wcp[kb * 6 + r] = (T)gs[r];
    }
  } else {
    // Point3 Difference: e = s (X - t), J = diag(s):  phi = - sum_r s_r^2 q_r (X_r - t_r)
    const int k = c - s.num_obs - s.num_cam_priors, pt_i = s.pt_prior_pt[k];
    const T* Xp = static_cast<const T*>(d.points) + ((int64_t)pt_i * B + b) * 3;
    const T* tp = static_cast<const T*>(d.pt_prior_target) + ((int64_t)k * (d.pt_prior_target_bstride ? B : 1)) * 3 +
                  (int64_t)b * d.pt_prior_target_bstride;
    const T* wp = static_cast<const T*>(d.w_pt_prior) + ((int64_t)k * (d.w_pt_prior_bstride ? B : 1)) * 3 +
                  (int64_t)b * d.w_pt_prior_bstride;
    const int64_t kb = (int64_t)k * B + b;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double q = (double)wv[nc + 3 * pt_i + r], sr = (double)wp[r], df = (double)Xp[r] - (double)tp[r];
      if (g_ppt) g_ppt[kb * 3 + r] = (T)(sr * sr * q);
      if (g_wpp) g_wpp[kb * 3 + r] = (T)(-2.0 * sr * q * df);
    }
  }
}

}  // namespace thx

using namespace thx;

extern "C" {

int thx_ba_vjp(const thx_ba_structure* s, const thx_ba_data* d, const void* w, int64_t ldw, void* grad_feat, void* grad_w_obs,
               void* grad_focal, void* grad_k1, void* grad_k2, void* grad_log_radius_obs, void* grad_cam_prior_target,
               void* grad_w_cam_prior, void* grad_pt_prior_target, void* grad_w_pt_prior, int dtype, const thx_lie_eps* eps,
               void* stream) {
  if (!s || !d || !w || !eps) return fail("thx_ba_vjp: null argument");
  if (ldw < 6 * (int64_t)s->num_cams + 3 * (int64_t)s->num_points) return fail("thx_ba_vjp: ldw < n");
  if ((grad_focal != nullptr) != (grad_k1 != nullptr) || (grad_focal != nullptr) != (grad_k2 != nullptr))
    return fail("thx_ba_vjp: grad_focal / grad_k1 / grad_k2 come together");
  if (d->robust_obs && !d->log_radius_obs) return fail("thx_ba_vjp: robust cost without log_loss_radius");
  dim3 grid((d->batch + 63) / 64, s->num_obs + s->num_cam_priors + s->num_pt_priors), block(64);
  if (grid.y == 0) return 0;
  const Eps<double> e = dtype == THX_F32 ? Eps<double>{(double)(float)eps->near_zero, (double)(float)eps->d_near_zero, (double)(float)eps->near_pi}
                                         : Eps<double>{eps->near_zero, eps->d_near_zero, eps->near_pi};
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(ba_vjp_kernel<float>, grid, block, 0, as_stream(stream), *s, *d, (const float*)w, ldw,
                                  (float*)grad_feat, (float*)grad_w_obs, (float*)grad_focal, (float*)grad_k1, (float*)grad_k2,
                                  (float*)grad_log_radius_obs, (float*)grad_cam_prior_target, (float*)grad_w_cam_prior,
                                  (float*)grad_pt_prior_target, (float*)grad_w_pt_prior, e),
               hipLaunchKernelGGL(ba_vjp_kernel<double>, grid, block, 0, as_stream(stream), *s, *d, (const double*)w, ldw,
                                  (double*)grad_feat, (double*)grad_w_obs, (double*)grad_focal, (double*)grad_k1, (double*)grad_k2,
                                  (double*)grad_log_radius_obs, (double*)grad_cam_prior_target, (double*)grad_w_cam_prior,
                                  (double*)grad_pt_prior_target, (double*)grad_w_pt_prior, e));
  return check_launch("thx_ba_vjp");
}

}  // extern "C"
