// Bundle adjustment on the dense GN/LM path (include/theseus_hip.h, "Bundle adjustment"): fused Reprojection residual /
// Jacobian evaluation, block assembly, Schur complement on the camera block, back substitution, error metric.
// Same mapping as the pose-graph kernels: one lane per (entity, problem), batch index fastest across the wave, entity
// tables wave-uniform, owner-computes (no atomics, bit-reproducible); per-cost arithmetic in fp64 registers.
#include "common.cuh"
#include "robust.cuh"

namespace thx {

struct Reproj {
  double Jc[12];  // 2x6 row major (weighted)
  double Jp[6];   // 2x3
  double e[2];
};

// theseus/embodied/measurements/reprojection.py:54-94 (+ se3_impl.py:757-777), rows scaled by w (cost_weight.py:81-136)
__device__ __forceinline__ void reproj_eval(const SE3<double>& cam, const double* X, const double* feat, double f, double k1,
                                            double k2, const double* w, bool want_jac, Reproj& r) {
  double pc[3];
  mat3_vec(cam.R, X, pc);
#pragma unroll
  for (int i = 0; i < 3; ++i) pc[i] += cam.t[i];
  const double iz = 1.0 / pc[2];
  const double proj[2] = {-pc[0] * iz, -pc[1] * iz};
  const double q = proj[0] * proj[0] + proj[1] * proj[1];
  const double factor = f * (1.0 + q * (k1 + q * k2));
  r.e[0] = (proj[0] * factor - feat[0]) * w[0];
  r.e[1] = (proj[1] * factor - feat[1]) * w[1];
  if (!want_jac) return;
  const double dfactor = f * (k1 + 2.0 * q * k2);
  // J = [R, -R hat(X) | R]  (3 x 9)
  double J[27];
  const double hx[9] = {0.0, -X[2], X[1], X[2], 0.0, -X[0], -X[1], X[0], 0.0};
  double RH[9];
  mat3_mul(cam.R, hx, RH);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      J[9 * i + j] = cam.R[3 * i + j];
      J[9 * i + 3 + j] = -RH[3 * i + j];
      J[9 * i + 6 + j] = cam.R[3 * i + j];
    }
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const double j2 = J[18 + j] * iz;
    const double pj0 = (pc[0] * j2 - J[j]) * iz;        // derivative of N/D is (N' - N D'/D) / D, with the sign of proj
    const double pj1 = (pc[1] * j2 - J[9 + j]) * iz;
    const double qj = 2.0 * (proj[0] * pj0 + proj[1] * pj1);
    const double o0 = (pj0 * factor + proj[0] * qj * dfactor) * w[0];
    const double o1 = (pj1 * factor + proj[1] * qj * dfactor) * w[1];
    if (j < 6) {
      r.Jc[j] = o0;
      r.Jc[6 + j] = o1;
    } else {
      r.Jp[j - 6] = o0;
      r.Jp[3 + j - 6] = o1;
    }
  }
}

template <typename T>
__device__ __forceinline__ SE3<double> load_cam(const T* __restrict__ p) {
  SE3<double> X;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    X.R[3 * i] = (double)p[4 * i];
    X.R[3 * i + 1] = (double)p[4 * i + 1];
    X.R[3 * i + 2] = (double)p[4 * i + 2];
    X.t[i] = (double)p[4 * i + 3];
  }
  return X;
}

// The fp64 block workspaces of this path -- W (O, 18), Hcc (C, 36), Hpp / Hinv (Np, 6), tvec (Np, 3) -- are PLANAR: (entity,
// component, B) with the batch index innermost.  A lane is one (entity, problem); the 64 lanes of a wave are 64 consecutive
// problems of ONE entity, so component i of all lanes is 512 contiguous bytes: every load / store instruction moves four full
// 128-B lines.  With the record-per-lane layout (entity, B, component) a wave's 16-byte load of an 144-byte W record touched 72
// lines and every line was touched by nine different instructions (ba_schur_block: 2.4-2.7 ms for 1.3 GB of W).
#ifndef THX_BA_PLANAR
#define THX_BA_PLANAR 1
#endif
template <int NC>
__device__ __forceinline__ int64_t ws_at(int64_t entity, int i, int b, int B) {
  return THX_BA_PLANAR ? (entity * NC + i) * (int64_t)B + b : (entity * (int64_t)B + b) * NC + i;
}

// everything one observation needs besides its two variables
template <typename T>
struct ObsAux {
  double feat[2], w[2], f, k1, k2, lr;
};
template <typename T>
__device__ __forceinline__ void load_obs(const thx_ba_data& d, int o, int c, int b, ObsAux<T>& a) {
  const int B = d.batch;
  const T* fp = static_cast<const T*>(d.feat) + ((int64_t)o * (d.feat_bstride ? B : 1)) * 2 + (int64_t)b * d.feat_bstride;
  const T* wp = static_cast<const T*>(d.w_obs) + ((int64_t)o * (d.w_obs_bstride ? B : 1)) * 2 + (int64_t)b * d.w_obs_bstride;
  const int64_t ci = (int64_t)c * (d.calib_bstride ? B : 1) + (int64_t)b * d.calib_bstride;
  a.feat[0] = (double)fp[0]; a.feat[1] = (double)fp[1];
  a.w[0] = (double)wp[0]; a.w[1] = (double)wp[1];
  a.f = (double)static_cast<const T*>(d.focal)[ci];
  a.k1 = (double)static_cast<const T*>(d.k1)[ci];
  a.k2 = (double)static_cast<const T*>(d.k2)[ci];
  a.lr = d.robust_obs ? load_log_radius<T>(d.log_radius_obs, o, b, B, d.log_radius_obs_bstride) : 0.0;
}
__device__ __forceinline__ void robustify_obs(int code, double lr, Reproj& r, bool jac) {
  if (code == THX_LOSS_NONE) return;
  double f[2];
  robust_row_scale<2>(code, r.e, lr, f);   // robust.cuh: one factor, or one per row with THX_LOSS_FLATTEN
  r.e[0] *= f[0]; r.e[1] *= f[1];
  if (jac) {
#pragma unroll
    for (int i = 0; i < 6; ++i) { r.Jc[i] *= f[0]; r.Jc[6 + i] *= f[1]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { r.Jp[i] *= f[0]; r.Jp[3 + i] *= f[1]; }
  }
}

// ---- pass A: one lane per (point, problem): Hpp, gp, W = Jc^T Jp of the point's observations ----
template <typename T>
__global__ void __launch_bounds__(64)
ba_point_kernel(thx_ba_structure s, thx_ba_data d, double* __restrict__ Hpp, double* __restrict__ W,
                double* __restrict__ gd, T* __restrict__ g, T* __restrict__ diag, int64_t ldv) {
  const int b = blockIdx.x * 64 + threadIdx.x, p = blockIdx.y, B = d.batch;
  if (b >= B) return;
  const T* Xp = static_cast<const T*>(d.points) + ((int64_t)p * B + b) * 3;
  const double X[3] = {(double)Xp[0], (double)Xp[1], (double)Xp[2]};
  double h[6] = {0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0};
  for (int k = s.pt_ptr[p]; k < s.pt_ptr[p + 1]; ++k) {
    const int o = s.pt_obs[k], c = s.obs_cam[o];
    const SE3<double> cam = load_cam(static_cast<const T*>(d.cams) + ((int64_t)c * B + b) * 12);
    ObsAux<T> a;
    load_obs<T>(d, o, c, b, a);
    Reproj r;
    reproj_eval(cam, X, a.feat, a.f, a.k1, a.k2, a.w, true, r);
    robustify_obs(d.robust_obs, a.lr, r, true);
    h[0] += r.Jp[0] * r.Jp[0] + r.Jp[3] * r.Jp[3];
    h[1] += r.Jp[0] * r.Jp[1] + r.Jp[3] * r.Jp[4];
    h[2] += r.Jp[0] * r.Jp[2] + r.Jp[3] * r.Jp[5];
    h[3] += r.Jp[1] * r.Jp[1] + r.Jp[4] * r.Jp[4];
    h[4] += r.Jp[1] * r.Jp[2] + r.Jp[4] * r.Jp[5];
    h[5] += r.Jp[2] * r.Jp[2] + r.Jp[5] * r.Jp[5];
#pragma unroll
    for (int i = 0; i < 3; ++i) gp[i] -= r.Jp[i] * r.e[0] + r.Jp[3 + i] * r.e[1];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double wv = r.Jc[i] * r.Jp[j] + r.Jc[6 + i] * r.Jp[3 + j];
#ifdef THX_EXP_W32   // numerics experiment: what an fp32 W workspace would do to fp32 problems (rounded on store, layout unchanged)
        if (sizeof(T) == 4) wv = (double)(float)wv;
#endif
        W[ws_at<18>(o, 3 * i + j, b, B)] = wv;
      }
  }
  for (int k = s.pt_prior_ptr[p]; k < s.pt_prior_ptr[p + 1]; ++k) {
    const int id = s.pt_prior_id[k];
    const T* tg = static_cast<const T*>(d.pt_prior_target) + ((int64_t)id * (d.pt_prior_target_bstride ? B : 1)) * 3 +
                  (int64_t)b * d.pt_prior_target_bstride;
    const T* wp = static_cast<const T*>(d.w_pt_prior) + ((int64_t)id * (d.w_pt_prior_bstride ? B : 1)) * 3 +
                  (int64_t)b * d.w_pt_prior_bstride;
    const int dgi[3] = {0, 3, 5};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double w = (double)wp[i], e = (X[i] - (double)tg[i]) * w;
      h[dgi[i]] += w * w;
      gp[i] -= w * e;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) Hpp[ws_at<6>(p, i, b, B)] = h[i];
  const int64_t col = 6 * (int64_t)s.num_cams + 3 * p;
  T* gb = g + (int64_t)b * ldv + col;
  T* db = diag + (int64_t)b * ldv + col;
  double* gdb = gd + (int64_t)b * ldv + col;
  gdb[0] = gp[0]; gdb[1] = gp[1]; gdb[2] = gp[2];
  gb[0] = (T)gp[0]; gb[1] = (T)gp[1]; gb[2] = (T)gp[2];
  db[0] = (T)h[0]; db[1] = (T)h[3]; db[2] = (T)h[5];
}

// ---- pass B: one lane per (camera, problem): Hcc, gc ----
template <typename T>
__global__ void __launch_bounds__(64)
ba_camera_kernel(thx_ba_structure s, thx_ba_data d, double* __restrict__ Hcc, double* __restrict__ gd, T* __restrict__ g,
                 T* __restrict__ diag, int64_t ldv, Eps<T> eps) {
  const int b = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y, B = d.batch;
  if (b >= B) return;
  const T* cp = static_cast<const T*>(d.cams) + ((int64_t)c * B + b) * 12;
  const SE3<double> cam = load_cam(cp);
  double Hc[36], gc[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) Hc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) gc[i] = 0.0;
  for (int k = s.cam_ptr[c]; k < s.cam_ptr[c + 1]; ++k) {
    const int o = s.cam_obs[k], p = s.obs_pt[o];
    const T* Xp = static_cast<const T*>(d.points) + ((int64_t)p * B + b) * 3;
    const double X[3] = {(double)Xp[0], (double)Xp[1], (double)Xp[2]};
    ObsAux<T> a;
    load_obs<T>(d, o, c, b, a);
    Reproj r;
    reproj_eval(cam, X, a.feat, a.f, a.k1, a.k2, a.w, true, r);
    robustify_obs(d.robust_obs, a.lr, r, true);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int j = 0; j < 6; ++j) Hc[6 * i + j] += r.Jc[i] * r.Jc[j] + r.Jc[6 + i] * r.Jc[6 + j];
      gc[i] -= r.Jc[i] * r.e[0] + r.Jc[6 + i] * r.e[1];
    }
  }
  for (int k = s.cam_prior_ptr[c]; k < s.cam_prior_ptr[c + 1]; ++k) {
    const int id = s.cam_prior_id[k];
    const SE3<double> Tg = load_cam(static_cast<const T*>(d.cam_prior_target) +
                                    ((int64_t)id * (d.cam_prior_target_bstride ? B : 1)) * 12 + (int64_t)b * d.cam_prior_target_bstride);
    const T* wp = static_cast<const T*>(d.w_cam_prior) + ((int64_t)id * (d.w_cam_prior_bstride ? B : 1)) * 6 +
                  (int64_t)b * d.w_cam_prior_bstride;
    double w[6], ev[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = (double)wp[i];
    SJac<double> Jd;
    local_eval<double>(Tg, cam, w, widen(eps), ev, &Jd, true);
    sjac_tmul_acc(Jd, Jd, Hc);
    sjac_tvec_sub(Jd, ev, gc);
  }
#pragma unroll
  for (int i = 0; i < 36; ++i) Hcc[ws_at<36>(c, i, b, B)] = Hc[i];
  T* gb = g + (int64_t)b * ldv + 6 * c;
  T* db = diag + (int64_t)b * ldv + 6 * c;
  double* gdb = gd + (int64_t)b * ldv + 6 * c;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    gdb[i] = gc[i];
    gb[i] = (T)gc[i];
    db[i] = (T)Hc[7 * i];
  }
}

// ---- Schur 1: damped point blocks inverted, t = Hpp'^-1 gp ----
template <typename T>
__global__ void __launch_bounds__(64)
ba_point_invert_kernel(thx_ba_structure s, int B, const double* __restrict__ Hpp, const double* __restrict__ g, int64_t ldv,
                       const T* __restrict__ damping, int ellipsoidal, T eps, double* __restrict__ Hinv,
                       double* __restrict__ tvec, int32_t* __restrict__ info) {
  const int b = blockIdx.x * 64 + threadIdx.x, p = blockIdx.y;
  if (b >= B) return;
  double Hp[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) Hp[i] = Hpp[ws_at<6>(p, i, b, B)];
  double hd[3] = {Hp[0], Hp[3], Hp[5]};
  if (damping) {  // DenseSolver._apply_damping (dense_solver.py:38-64)
    const double lam = (double)damping[b];
#pragma unroll
    for (int i = 0; i < 3; ++i) hd[i] = ellipsoidal ? hd[i] + (lam * hd[i] + (double)eps) : hd[i] + lam;
  }
  const double a = hd[0], bb = Hp[1], c = Hp[2], dd = hd[1], e = Hp[4], f = hd[2];
  // adjugate of the symmetric 3x3 [[a,b,c],[b,d,e],[c,e,f]]
  const double A = dd * f - e * e, Bc = c * e - bb * f, C = bb * e - c * dd;
  const double det = a * A + bb * Bc + c * C;
  const double m2 = a * dd - bb * bb;
  if (!(a > 0.0) || !(m2 > 0.0) || !(det > 0.0)) info[b] = 6 * s.num_cams + 3 * p + 1;  // any writer: value only flags failure
  const double id = 1.0 / det;
  const double inv[6] = {A * id, Bc * id, C * id, (a * f - c * c) * id, (bb * c - a * e) * id, m2 * id};
#pragma unroll
  for (int i = 0; i < 6; ++i) Hinv[ws_at<6>(p, i, b, B)] = inv[i];
  const double* gp = g + (int64_t)b * ldv + 6 * (int64_t)s.num_cams + 3 * p;
  const double g0 = gp[0], g1 = gp[1], g2 = gp[2];
  tvec[ws_at<3>(p, 0, b, B)] = inv[0] * g0 + inv[1] * g1 + inv[2] * g2;
  tvec[ws_at<3>(p, 1, b, B)] = inv[1] * g0 + inv[3] * g1 + inv[4] * g2;
  tvec[ws_at<3>(p, 2, b, B)] = inv[2] * g0 + inv[4] * g1 + inv[5] * g2;
}

// one 6-element row of an S block: three 2-element stores when the frame allows (ld even: every row of a 6 x 6 block then
// starts on a 2-element boundary) -- a lane's row is 24 (fp32) / 48 (fp64) contiguous bytes of ITS problem's frame, nothing
// coalesces across lanes, so the number of store instructions (each 64 partial-line writes) is what counts.  Measured on one box
// (profiles/r3/y_ab_ba_schur_store_modes.txt): nontemporal stores 2.3x SLOWER (the write-back L2 merges the segments of a line),
// one 4- + one 2-element store per row slower too (3.6 vs 3.15 ms: the alignment branch)
template <typename T>
__device__ __forceinline__ void store_row6(T* __restrict__ p, const double* v, bool pairs) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  if (pairs) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T2 w = {(T)v[2 * c], (T)v[2 * c + 1]};
      *reinterpret_cast<T2*>(p + 2 * c) = w;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 6; ++c) p[c] = (T)v[c];
  }
}

// a 6 x 6 block as 36 contiguous values of a block list (thx_ba_schur_blocks): 144 (fp32) / 288 (fp64) contiguous bytes per lane in
// 16-byte stores -- every store instruction writes whole 16-byte segments, against six 24-byte row segments in a dense frame (which
// the memory system saw as 2.85 GB of writes for 1.4 GB of blocks).  ``transposed``: the list holds the block of the pair the other
// way round (the solver's elimination order).
template <typename T>
__device__ __forceinline__ void store_block36(T* __restrict__ p, const double* v, bool transposed) {
  constexpr int VEC = 16 / sizeof(T);
  typedef T TV __attribute__((ext_vector_type(VEC)));
  TV* dst = reinterpret_cast<TV*>(p);
  if (transposed) {
#pragma unroll
    for (int k = 0; k < 36 / VEC; ++k) {
      TV w;
#pragma unroll
      for (int e = 0; e < VEC; ++e) w[e] = (T)v[6 * ((VEC * k + e) % 6) + (VEC * k + e) / 6];
      dst[k] = w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 36 / VEC; ++k) {
      TV w;
#pragma unroll
      for (int e = 0; e < VEC; ++e) w[e] = (T)v[VEC * k + e];
      dst[k] = w;
    }
  }
}

// where thx_ba_schur_blocks puts the blocks of S (diag_blk == nullptr: the dense frame of thx_ba_schur)
struct SchurBlockDst {
  const int32_t* diag_blk;   // (C): block id of S_cc; nullptr: the dense frame of thx_ba_schur
  const int32_t* blk_dst;    // (num_blocks): block id of the k-th camera pair | bit 30: stored transposed
  int64_t bstride;           // elements per problem
};

// M (6x3) = W (6x3) * Hinv (sym 3x3)
__device__ __forceinline__ void w_times_sym(const double* W, const double* h, double* M) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double w0 = W[3 * i], w1 = W[3 * i + 1], w2 = W[3 * i + 2];
    M[3 * i] = w0 * h[0] + w1 * h[1] + w2 * h[2];
    M[3 * i + 1] = w0 * h[1] + w1 * h[3] + w2 * h[4];
    M[3 * i + 2] = w0 * h[2] + w1 * h[4] + w2 * h[5];
  }
}

// ---- Schur 2a: one lane per (off-diagonal block (c1, c2 < c1), problem): S_c1c2 = -sum_pairs W_o1 Hpp'^-1 W_o2^T.
//      (One lane per block ROW left 8 waves per CU walking ~175 pairs each: 3.8 ms at 512 cameras / 32768 observations;
//      per block there are ~40x more lanes with a handful of pairs each.)
template <typename T>
__global__ void __launch_bounds__(64)
ba_schur_block_kernel(thx_ba_structure s, int B, const double* __restrict__ W, const double* __restrict__ Hinv,
                      T* __restrict__ S, int64_t ld, SchurBlockDst bl) {
  const int b = blockIdx.y * 64 + threadIdx.x, k = blockIdx.x;  // blocks along x: their number can exceed 65535
  if (b >= B) return;
  const int c1 = s.blk_c1[k], c2 = s.blk_c2[k];
  double Off[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) Off[i] = 0.0;
  for (int q = s.blk_ptr[2 * k]; q < s.blk_ptr[2 * k + 1]; ++q) {
    const int o1 = s.pair_o1[q], o2 = s.pair_o2[q];
    const int p = s.obs_pt[o1];
    double W1[18], W2[18], h[6], M[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) { W1[i] = W[ws_at<18>(o1, i, b, B)]; W2[i] = W[ws_at<18>(o2, i, b, B)]; }
#pragma unroll
    for (int i = 0; i < 6; ++i) h[i] = Hinv[ws_at<6>(p, i, b, B)];
    w_times_sym(W1, h, M);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c)
        Off[6 * r + c] -= M[3 * r] * W2[3 * c] + M[3 * r + 1] * W2[3 * c + 1] + M[3 * r + 2] * W2[3 * c + 2];
  }
  if (bl.diag_blk) {   // (block list; block uniform)
    const int d = bl.blk_dst[k];
    store_block36(S + (int64_t)b * bl.bstride + (int64_t)(d & 0x3fffffff) * 36, Off, ((d >> 30) & 1) != 0);
    return;
  }
  T* Sb = S + (int64_t)b * ld * ld;
  const bool pairs = (ld & 1) == 0;
#pragma unroll
  for (int r = 0; r < 6; ++r) store_row6(Sb + (int64_t)(6 * c1 + r) * ld + 6 * c2, Off + 6 * r, pairs);
}

// ---- Schur 2b: one lane per (camera c1, problem): the diagonal block S_c1c1 (Hcc' - its pairs) and rhs_c1 ----
template <typename T>
__global__ void __launch_bounds__(64)
ba_schur_kernel(thx_ba_structure s, int B, const double* __restrict__ Hcc, const double* __restrict__ W,
                const double* __restrict__ g, int64_t ldv, const T* __restrict__ damping, int ellipsoidal, T eps,
                const double* __restrict__ Hinv, const double* __restrict__ tvec, T* __restrict__ S, int64_t ld,
                T* __restrict__ rhs, int64_t ldr, SchurBlockDst bl) {
  const int b = blockIdx.x * 64 + threadIdx.x, c1 = blockIdx.y;
  if (b >= B) return;
  double Dg[36], rv[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) Dg[i] = Hcc[ws_at<36>(c1, i, b, B)];
  if (damping) {
    const double lam = (double)damping[b];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double h = Dg[7 * i];
      Dg[7 * i] = ellipsoidal ? h + (lam * h + (double)eps) : h + lam;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) rv[i] = g[(int64_t)b * ldv + 6 * c1 + i];
  // rhs -= sum_o W_o t_p(o): every observation o of the camera is its own diagonal pair (o, o) -- done inside the pair loop
  // below, on the W block that loop has loaded anyway
  T* Sb = S + (int64_t)b * ld * ld;
  for (int k = s.pair_dptr[c1]; k < s.pair_ptr[c1 + 1]; ++k) {  // the pairs with cam(o2) = c1
    const int o1 = s.pair_o1[k], o2 = s.pair_o2[k];
    const int p = s.obs_pt[o1];
    double W1[18], W2[18], h[6], M[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) W1[i] = W[ws_at<18>(o1, i, b, B)];
    if (o1 == o2) {   // (block uniform: the structure is shared by the batch) a camera sees a point once -- the usual case
#pragma unroll
      for (int i = 0; i < 18; ++i) W2[i] = W1[i];
      const double t0 = tvec[ws_at<3>(p, 0, b, B)], t1 = tvec[ws_at<3>(p, 1, b, B)], t2 = tvec[ws_at<3>(p, 2, b, B)];
#pragma unroll
      for (int i = 0; i < 6; ++i) rv[i] -= W1[3 * i] * t0 + W1[3 * i + 1] * t1 + W1[3 * i + 2] * t2;
    } else {
#pragma unroll
      for (int i = 0; i < 18; ++i) W2[i] = W[ws_at<18>(o2, i, b, B)];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) h[i] = Hinv[ws_at<6>(p, i, b, B)];
    w_times_sym(W1, h, M);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c)
        Dg[6 * r + c] -= M[3 * r] * W2[3 * c] + M[3 * r + 1] * W2[3 * c + 1] + M[3 * r + 2] * W2[3 * c + 2];
  }
  if (bl.diag_blk) {
    store_block36(S + (int64_t)b * bl.bstride + (int64_t)bl.diag_blk[c1] * 36, Dg, false);
  } else {
    const bool pairs = (ld & 1) == 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) store_row6(Sb + (int64_t)(6 * c1 + r) * ld + 6 * c1, Dg + 6 * r, pairs);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) rhs[(int64_t)b * ldr + 6 * c1 + i] = (T)rv[i];
}

// ---- back substitution: delta_p = t_p - Hinv_p sum_o W_o^T delta_c(o) ----
template <typename T>
__global__ void __launch_bounds__(64)
ba_backsub_kernel(thx_ba_structure s, int B, const double* __restrict__ W, const double* __restrict__ Hinv,
                  const double* __restrict__ tvec, T* __restrict__ delta, int64_t ldv) {
  const int b = blockIdx.x * 64 + threadIdx.x, p = blockIdx.y;
  if (b >= B) return;
  double acc[3] = {0, 0, 0};
  for (int k = s.pt_ptr[p]; k < s.pt_ptr[p + 1]; ++k) {
    const int o = s.pt_obs[k], c = s.obs_cam[o];
    const T* dc = delta + (int64_t)b * ldv + 6 * c;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double di = (double)dc[i];
      acc[0] += W[ws_at<18>(o, 3 * i, b, B)] * di;
      acc[1] += W[ws_at<18>(o, 3 * i + 1, b, B)] * di;
      acc[2] += W[ws_at<18>(o, 3 * i + 2, b, B)] * di;
    }
  }
  double h[6], tp[3];
#pragma unroll
  for (int i = 0; i < 6; ++i) h[i] = Hinv[ws_at<6>(p, i, b, B)];
#pragma unroll
  for (int i = 0; i < 3; ++i) tp[i] = tvec[ws_at<3>(p, i, b, B)];
  T* dp = delta + (int64_t)b * ldv + 6 * (int64_t)s.num_cams + 3 * p;
  dp[0] = (T)(tp[0] - (h[0] * acc[0] + h[1] * acc[1] + h[2] * acc[2]));
  dp[1] = (T)(tp[1] - (h[1] * acc[0] + h[3] * acc[1] + h[4] * acc[2]));
  dp[2] = (T)(tp[2] - (h[2] * acc[0] + h[4] * acc[1] + h[5] * acc[2]));
}

// ---- error metric: BA_ERR_CHUNKS partial sums per problem (the batch is small, the cost count large) ----
constexpr int BA_ERR_CHUNKS = THX_BA_ERR_CHUNKS;
template <typename T>
__global__ void __launch_bounds__(64)
ba_error_partial_kernel(thx_ba_structure s, thx_ba_data d, T* __restrict__ partials, Eps<T> eps) {
  const int b = blockIdx.x * 64 + threadIdx.x, ch = blockIdx.y, B = d.batch;
  if (b >= B) return;
  double acc = 0.0;
  auto range = [&](int n, int& lo, int& hi) __attribute__((always_inline)) {
    const int per = (n + BA_ERR_CHUNKS - 1) / BA_ERR_CHUNKS;
    lo = ch * per;
    hi = min(n, lo + per);
  };
  int lo, hi;
  range(s.num_obs, lo, hi);
  for (int o = lo; o < hi; ++o) {
    const int c = s.obs_cam[o], p = s.obs_pt[o];
    const SE3<double> cam = load_cam(static_cast<const T*>(d.cams) + ((int64_t)c * B + b) * 12);
    const T* Xp = static_cast<const T*>(d.points) + ((int64_t)p * B + b) * 3;
    const double X[3] = {(double)Xp[0], (double)Xp[1], (double)Xp[2]};
    ObsAux<T> a;
    load_obs<T>(d, o, c, b, a);
    Reproj r;
    reproj_eval(cam, X, a.feat, a.f, a.k1, a.k2, a.w, false, r);
    acc += robust_sq_error<2>(d.robust_obs, r.e, a.lr);
  }
  range(s.num_cam_priors, lo, hi);
  for (int k = lo; k < hi; ++k) {
    const int c = s.cam_prior_cam[k];
    const SE3<double> Xc = load_cam(static_cast<const T*>(d.cams) + ((int64_t)c * B + b) * 12);
    const SE3<double> Tg = load_cam(static_cast<const T*>(d.cam_prior_target) +
                                    ((int64_t)k * (d.cam_prior_target_bstride ? B : 1)) * 12 + (int64_t)b * d.cam_prior_target_bstride);
    const T* wp = static_cast<const T*>(d.w_cam_prior) + ((int64_t)k * (d.w_cam_prior_bstride ? B : 1)) * 6 +
                  (int64_t)b * d.w_cam_prior_bstride;
    double w[6], ev[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = (double)wp[i];
    local_eval<double>(Tg, Xc, w, widen(eps), ev, nullptr, false);
#pragma unroll
    for (int i = 0; i < 6; ++i) acc += ev[i] * ev[i];
  }
  range(s.num_pt_priors, lo, hi);
  for (int k = lo; k < hi; ++k) {
    const int p = s.pt_prior_pt[k];
    const T* Xp = static_cast<const T*>(d.points) + ((int64_t)p * B + b) * 3;
    const T* tg = static_cast<const T*>(d.pt_prior_target) + ((int64_t)k * (d.pt_prior_target_bstride ? B : 1)) * 3 +
                  (int64_t)b * d.pt_prior_target_bstride;
    const T* wp = static_cast<const T*>(d.w_pt_prior) + ((int64_t)k * (d.w_pt_prior_bstride ? B : 1)) * 3 +
                  (int64_t)b * d.w_pt_prior_bstride;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double e = ((double)Xp[i] - (double)tg[i]) * (double)wp[i];
      acc += e * e;
    }
  }
  partials[(int64_t)ch * B + b] = (T)acc;
}
template <typename T>
__global__ void ba_error_reduce_kernel(const T* __restrict__ partials, T* __restrict__ err, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double acc = 0.0;  // 256 partials: accumulate in fp64 so that the sum carries one fp32 rounding, not 256
  for (int c = 0; c < BA_ERR_CHUNKS; ++c) acc += (double)partials[(int64_t)c * B + b];
  err[b] = (T)(0.5 * acc);
}

template <typename T>
__global__ void __launch_bounds__(64)
vec_retract_kernel(const T* __restrict__ x, const T* __restrict__ delta, int64_t ldd, int64_t col0, T step,
                   const uint8_t* __restrict__ ignore, T* __restrict__ out, int N, int dof, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x, p = blockIdx.y;
  if (b >= B) return;
  const bool keep = ignore && ignore[b];
  for (int i = 0; i < dof; ++i) {
    const int64_t o = ((int64_t)p * B + b) * dof + i;
    out[o] = keep ? x[o] : x[o] + delta[(int64_t)b * ldd + col0 + (int64_t)p * dof + i] * step;
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
lm_accept_diag_kernel(const T* __restrict__ delta, const T* __restrict__ g, const T* __restrict__ diag, int64_t ldv, int n,
                      T* __restrict__ damping, const T* __restrict__ prev_err, const T* __restrict__ new_err,
                      int ellipsoidal, T accept, T down, T up, uint8_t* __restrict__ reject) {
  // levenberg_marquardt.py:173-201 (same arithmetic as lm_accept_kernel, diag(H) read from a vector).  n is 25 k at the
  // bundle-adjustment size: four waves per problem, four independent element groups in flight per thread (one wave with one
  // element per trip was 0.31 ms at batch 256 -- latency, not bytes)
  __shared__ T part[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const T lam = damping[b];
  const T* db = delta + (int64_t)b * ldv;
  const T* gb = g + (int64_t)b * ldv;
  const T* hb = diag + (int64_t)b * ldv;
  T s4[4] = {T(0), T(0), T(0), T(0)};
  int i = tid;
  for (; i + 768 < n; i += 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const T dl = db[i + 256 * u];
      const T D = ellipsoidal ? hb[i + 256 * u] * lam : lam;
      s4[u] += dl * (D * dl + gb[i + 256 * u]);
    }
  }
  for (; i < n; i += 256) {
    const T dl = db[i];
    const T D = ellipsoidal ? hb[i] * lam : lam;
    s4[0] += dl * (D * dl + gb[i]);
  }
  T s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) part[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    s = (part[0] + part[1]) + (part[2] + part[3]);
    const T den = s / T(2);
    const T rho = (prev_err[b] - new_err[b]) / den;
    const bool rej = rho <= accept;
    T nl = rej ? lam * up : lam / down;
    nl = nl < T(1.0e-7) ? T(1.0e-7) : (nl > T(1.0e7) ? T(1.0e7) : nl);
    damping[b] = nl;
    reject[b] = rej ? 1 : 0;
  }
}

// ---- A v : the product the reference takes with its dense Jacobian (dense_linearization.py:73-74; read by Dogleg /
//      TrustRegion, dogleg.py:66, trust_region.py:97), one lane per (cost, problem), Jacobian blocks recomputed on the fly.
//      Rows in cost ADD order (row tables from the host), output transposed (m, B): coalesced over the batch. ----
template <typename T>
__global__ void __launch_bounds__(64)
ba_av_kernel(thx_ba_structure s, thx_ba_data d, const T* __restrict__ v, int64_t ldv, const int32_t* __restrict__ obs_row,
             const int32_t* __restrict__ cam_prior_row, const int32_t* __restrict__ pt_prior_row, T* __restrict__ out_t,
             Eps<T> eps) {
  const int b = blockIdx.x * 64 + threadIdx.x, B = d.batch;
  int c = blockIdx.y;
  if (b >= B) return;
  const T* vb = v + (int64_t)b * ldv;
  const int64_t pcol = 6 * (int64_t)s.num_cams;
  if (c < s.num_obs) {
    const int o = c, cam_i = s.obs_cam[o], p = s.obs_pt[o];
    const SE3<double> cam = load_cam(static_cast<const T*>(d.cams) + ((int64_t)cam_i * B + b) * 12);
    const T* Xp = static_cast<const T*>(d.points) + ((int64_t)p * B + b) * 3;
    const double X[3] = {(double)Xp[0], (double)Xp[1], (double)Xp[2]};
    ObsAux<T> a;
    load_obs<T>(d, o, cam_i, b, a);
    Reproj r;
    reproj_eval(cam, X, a.feat, a.f, a.k1, a.k2, a.w, true, r);
    robustify_obs(d.robust_obs, a.lr, r, true);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) acc += r.Jc[6 * i + k] * (double)vb[6 * cam_i + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) acc += r.Jp[3 * i + k] * (double)vb[pcol + 3 * p + k];
      out_t[(int64_t)(obs_row[o] + i) * B + b] = (T)acc;
    }
    return;
  }
  c -= s.num_obs;
  if (c < s.num_cam_priors) {
    const int k = c, cam_i = s.cam_prior_cam[k];
    const SE3<double> Xc = load_cam(static_cast<const T*>(d.cams) + ((int64_t)cam_i * B + b) * 12);
    const SE3<double> Tg = load_cam(static_cast<const T*>(d.cam_prior_target) +
                                    ((int64_t)k * (d.cam_prior_target_bstride ? B : 1)) * 12 + (int64_t)b * d.cam_prior_target_bstride);
    const T* wp = static_cast<const T*>(d.w_cam_prior) + ((int64_t)k * (d.w_cam_prior_bstride ? B : 1)) * 6 +
                  (int64_t)b * d.w_cam_prior_bstride;
    double w[6], ev[6], x[6], y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      w[i] = (double)wp[i];
      x[i] = (double)vb[6 * cam_i + i];
    }
    SJac<double> J;
    local_eval<double>(Tg, Xc, w, widen(eps), ev, &J, true);
    // J = [[a, c], [0, d]] (3x3 blocks, rows already weighted)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      y[i] = J.a[3 * i] * x[0] + J.a[3 * i + 1] * x[1] + J.a[3 * i + 2] * x[2] + J.c[3 * i] * x[3] + J.c[3 * i + 1] * x[4] +
             J.c[3 * i + 2] * x[5];
      y[3 + i] = J.d[3 * i] * x[3] + J.d[3 * i + 1] * x[4] + J.d[3 * i + 2] * x[5];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) out_t[(int64_t)(cam_prior_row[k] + i) * B + b] = (T)y[i];
    return;
  }
  c -= s.num_cam_priors;
  {
    const int k = c, p = s.pt_prior_pt[k];
    const T* wp = static_cast<const T*>(d.w_pt_prior) + ((int64_t)k * (d.w_pt_prior_bstride ? B : 1)) * 3 +
                  (int64_t)b * d.w_pt_prior_bstride;
#pragma unroll
    for (int i = 0; i < 3; ++i) out_t[(int64_t)(pt_prior_row[k] + i) * B + b] = (T)((double)wp[i] * (double)vb[pcol + 3 * p + i]);
  }
}

static int check_ba(const thx_ba_structure* s, const thx_ba_data* d) {
  if (!s || !d) return fail("thx_ba: null structure/data");
  if (s->num_cams <= 0 || s->num_points <= 0 || d->batch <= 0) return fail("thx_ba: empty problem");
  if (!d->cams || !d->points) return fail("thx_ba: null variables");
  if (s->num_obs > 0 && (!d->feat || !d->w_obs || !d->focal || !d->k1 || !d->k2)) return fail("thx_ba: null observation data");
  if (!loss_code_valid(d->robust_obs) || (d->robust_obs && !d->log_radius_obs)) return fail("thx_ba: bad robust loss");
  if ((d->feat_bstride != 0 && d->feat_bstride != 2) || (d->w_obs_bstride != 0 && d->w_obs_bstride != 2) ||
      (d->calib_bstride != 0 && d->calib_bstride != 1) || (d->cam_prior_target_bstride != 0 && d->cam_prior_target_bstride != 12) ||
      (d->w_cam_prior_bstride != 0 && d->w_cam_prior_bstride != 6) || (d->pt_prior_target_bstride != 0 && d->pt_prior_target_bstride != 3) ||
      (d->w_pt_prior_bstride != 0 && d->w_pt_prior_bstride != 3))
    return fail("thx_ba: bad batch stride");
  return 0;
}

}  // namespace thx

using namespace thx;

#define BA_LAUNCH(KERNEL, GRID, BLOCK, ...)                                                                   \
  THX_DISPATCH(dtype, hipLaunchKernelGGL(KERNEL<float>, GRID, BLOCK, 0, as_stream(stream), __VA_ARGS__),      \
               hipLaunchKernelGGL(KERNEL<double>, GRID, BLOCK, 0, as_stream(stream), __VA_ARGS__))

extern "C" {

int thx_ba_assemble(const thx_ba_structure* s, const thx_ba_data* d, void* Hcc, void* Hpp, void* W, void* gd, void* g,
                    void* diag, int64_t ldv, int dtype, const thx_lie_eps* eps, void* stream) {
  if (int r = check_ba(s, d)) return r;
  if (!Hcc || !Hpp || !gd || !g || !diag || !eps || (s->num_obs > 0 && !W)) return fail("thx_ba_assemble: null output");
  if (ldv < 6 * (int64_t)s->num_cams + 3 * (int64_t)s->num_points) return fail("thx_ba_assemble: ldv < n");
  const dim3 block(64), gp((d->batch + 63) / 64, s->num_points), gc((d->batch + 63) / 64, s->num_cams);
  THX_DISPATCH(dtype,
               {
                 hipLaunchKernelGGL(ba_point_kernel<float>, gp, block, 0, as_stream(stream), *s, *d, (double*)Hpp, (double*)W,
                                    (double*)gd, (float*)g, (float*)diag, ldv);
                 hipLaunchKernelGGL(ba_camera_kernel<float>, gc, block, 0, as_stream(stream), *s, *d, (double*)Hcc,
                                    (double*)gd, (float*)g, (float*)diag, ldv, make_eps<float>(eps));
               },
               {
                 hipLaunchKernelGGL(ba_point_kernel<double>, gp, block, 0, as_stream(stream), *s, *d, (double*)Hpp,
                                    (double*)W, (double*)gd, (double*)g, (double*)diag, ldv);
                 hipLaunchKernelGGL(ba_camera_kernel<double>, gc, block, 0, as_stream(stream), *s, *d, (double*)Hcc,
                                    (double*)gd, (double*)g, (double*)diag, ldv, make_eps<double>(eps));
               });
  return check_launch("thx_ba_assemble");
}

static int ba_schur_impl(const thx_ba_structure* s, int32_t B, const void* Hcc, const void* Hpp, const void* W, const void* gd,
                         int64_t ldv, const void* damping, int ellipsoidal, double damping_eps, void* S, int64_t ld, void* rhs,
                         int64_t ldr, void* Hinv, void* tvec, int32_t* info, int dtype, void* stream, SchurBlockDst bl,
                         const char* what) {
  const dim3 block(64), gp((B + 63) / 64, s->num_points), gc((B + 63) / 64, s->num_cams);
  hipMemsetAsync(info, 0, sizeof(int32_t) * (size_t)B, as_stream(stream));
  THX_DISPATCH(dtype,
               {
                 hipLaunchKernelGGL(ba_point_invert_kernel<float>, gp, block, 0, as_stream(stream), *s, B, (const double*)Hpp,
                                    (const double*)gd, ldv, (const float*)damping, ellipsoidal, (float)damping_eps,
                                    (double*)Hinv, (double*)tvec, info);
                 if (s->num_blocks > 0)
                   hipLaunchKernelGGL(ba_schur_block_kernel<float>, dim3(s->num_blocks, (B + 63) / 64), block, 0,
                                      as_stream(stream), *s, B, (const double*)W, (const double*)Hinv, (float*)S, ld, bl);
                 hipLaunchKernelGGL(ba_schur_kernel<float>, gc, block, 0, as_stream(stream), *s, B, (const double*)Hcc,
                                    (const double*)W, (const double*)gd, ldv, (const float*)damping, ellipsoidal,
                                    (float)damping_eps, (const double*)Hinv, (const double*)tvec, (float*)S, ld, (float*)rhs, ldr, bl);
               },
               {
                 hipLaunchKernelGGL(ba_point_invert_kernel<double>, gp, block, 0, as_stream(stream), *s, B,
                                    (const double*)Hpp, (const double*)gd, ldv, (const double*)damping, ellipsoidal,
                                    damping_eps, (double*)Hinv, (double*)tvec, info);
                 if (s->num_blocks > 0)
                   hipLaunchKernelGGL(ba_schur_block_kernel<double>, dim3(s->num_blocks, (B + 63) / 64), block, 0,
                                      as_stream(stream), *s, B, (const double*)W, (const double*)Hinv, (double*)S, ld, bl);
                 hipLaunchKernelGGL(ba_schur_kernel<double>, gc, block, 0, as_stream(stream), *s, B, (const double*)Hcc,
                                    (const double*)W, (const double*)gd, ldv, (const double*)damping, ellipsoidal, damping_eps,
                                    (const double*)Hinv, (const double*)tvec, (double*)S, ld, (double*)rhs, ldr, bl);
               });
  return check_launch(what);
}

int thx_ba_schur(const thx_ba_structure* s, int32_t B, const void* Hcc, const void* Hpp, const void* W, const void* gd,
                 int64_t ldv, const void* damping, int ellipsoidal, double damping_eps, void* S, int64_t ld, void* rhs,
                 int64_t ldr, void* Hinv, void* tvec, int32_t* info, int dtype, void* stream) {
  if (!s || !Hcc || !Hpp || !gd || !S || !rhs || !Hinv || !tvec || !info || B <= 0) return fail("thx_ba_schur: null argument");
  if (ld < 6 * (int64_t)s->num_cams || ldr < 6 * (int64_t)s->num_cams) return fail("thx_ba_schur: ld < 6 C");
  return ba_schur_impl(s, B, Hcc, Hpp, W, gd, ldv, damping, ellipsoidal, damping_eps, S, ld, rhs, ldr, Hinv, tvec, info, dtype,
                       stream, SchurBlockDst{nullptr, nullptr, 0}, "thx_ba_schur");
}

int thx_ba_schur_blocks(const thx_ba_structure* s, int32_t B, const void* Hcc, const void* Hpp, const void* W, const void* gd,
                        int64_t ldv, const void* damping, int ellipsoidal, double damping_eps, void* Sc, int64_t bstride,
                        const int32_t* diag_blk, const int32_t* blk_dst, void* rhs, int64_t ldr, void* Hinv, void* tvec,
                        int32_t* info, int dtype, void* stream) {
  if (!s || !Hcc || !Hpp || !gd || !Sc || !rhs || !Hinv || !tvec || !info || !diag_blk || B <= 0)
    return fail("thx_ba_schur_blocks: null argument");
  if (s->num_blocks > 0 && !blk_dst) return fail("thx_ba_schur_blocks: blk_dst missing");
  if (bstride < 36 * ((int64_t)s->num_cams + s->num_blocks) || (bstride & 3) != 0 || ldr < 6 * (int64_t)s->num_cams)
    return fail("thx_ba_schur_blocks: bstride (36 elements per block, a multiple of 4) / ldr < 6 C");
  return ba_schur_impl(s, B, Hcc, Hpp, W, gd, ldv, damping, ellipsoidal, damping_eps, Sc, 0, rhs, ldr, Hinv, tvec, info, dtype,
                       stream, SchurBlockDst{diag_blk, blk_dst, bstride}, "thx_ba_schur_blocks");
}

int thx_ba_backsub(const thx_ba_structure* s, int32_t B, const void* W, const void* Hinv, const void* tvec, void* delta,
                   int64_t ldv, int dtype, void* stream) {
  if (!s || !Hinv || !tvec || !delta || B <= 0 || (s->num_obs > 0 && !W)) return fail("thx_ba_backsub: null argument");
  const dim3 block(64), gp((B + 63) / 64, s->num_points);
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(ba_backsub_kernel<float>, gp, block, 0, as_stream(stream), *s, B, (const double*)W,
                                  (const double*)Hinv, (const double*)tvec, (float*)delta, ldv),
               hipLaunchKernelGGL(ba_backsub_kernel<double>, gp, block, 0, as_stream(stream), *s, B, (const double*)W,
                                  (const double*)Hinv, (const double*)tvec, (double*)delta, ldv));
  return check_launch("thx_ba_backsub");
}

int thx_ba_error(const thx_ba_structure* s, const thx_ba_data* d, void* partials, void* err, int dtype,
                 const thx_lie_eps* eps, void* stream) {
  if (int r = check_ba(s, d)) return r;
  if (!partials || !err || !eps) return fail("thx_ba_error: null output");
  const dim3 block(64), grid((d->batch + 63) / 64, BA_ERR_CHUNKS);
  const int B = d->batch;
  THX_DISPATCH(dtype,
               {
                 hipLaunchKernelGGL(ba_error_partial_kernel<float>, grid, block, 0, as_stream(stream), *s, *d,
                                    (float*)partials, make_eps<float>(eps));
                 hipLaunchKernelGGL(ba_error_reduce_kernel<float>, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream),
                                    (const float*)partials, (float*)err, B);
               },
               {
                 hipLaunchKernelGGL(ba_error_partial_kernel<double>, grid, block, 0, as_stream(stream), *s, *d,
                                    (double*)partials, make_eps<double>(eps));
                 hipLaunchKernelGGL(ba_error_reduce_kernel<double>, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream),
                                    (const double*)partials, (double*)err, B);
               });
  return check_launch("thx_ba_error");
}

int thx_ba_av(const thx_ba_structure* s, const thx_ba_data* d, const void* v, int64_t ldv, const int32_t* obs_row,
              const int32_t* cam_prior_row, const int32_t* pt_prior_row, void* out_t, int dtype, const thx_lie_eps* eps,
              void* stream) {
  if (int r = check_ba(s, d)) return r;
  if (!v || !out_t || !eps) return fail("thx_ba_av: null argument");
  if ((s->num_obs > 0 && !obs_row) || (s->num_cam_priors > 0 && !cam_prior_row) || (s->num_pt_priors > 0 && !pt_prior_row))
    return fail("thx_ba_av: null row table");
  if (ldv < 6 * (int64_t)s->num_cams + 3 * (int64_t)s->num_points) return fail("thx_ba_av: ldv < n");
  const int costs = s->num_obs + s->num_cam_priors + s->num_pt_priors;
  if (costs == 0) return 0;
  const dim3 block(64), grid((d->batch + 63) / 64, costs);
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(ba_av_kernel<float>, grid, block, 0, as_stream(stream), *s, *d, (const float*)v, ldv, obs_row,
                                  cam_prior_row, pt_prior_row, (float*)out_t, make_eps<float>(eps)),
               hipLaunchKernelGGL(ba_av_kernel<double>, grid, block, 0, as_stream(stream), *s, *d, (const double*)v, ldv, obs_row,
                                  cam_prior_row, pt_prior_row, (double*)out_t, make_eps<double>(eps)));
  return check_launch("thx_ba_av");
}

int thx_vec_retract(const void* x, const void* delta, int64_t ldd, int64_t col0, double step, const uint8_t* ignore_mask,
                    void* out, int32_t N, int32_t dof, int32_t B, int dtype, void* stream) {
  if (!x || !delta || !out || N <= 0 || dof <= 0 || B <= 0) return fail("thx_vec_retract: bad arguments");
  const dim3 block(64), grid((B + 63) / 64, N);
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(vec_retract_kernel<float>, grid, block, 0, as_stream(stream), (const float*)x,
                                  (const float*)delta, ldd, col0, (float)step, ignore_mask, (float*)out, N, dof, B),
               hipLaunchKernelGGL(vec_retract_kernel<double>, grid, block, 0, as_stream(stream), (const double*)x,
                                  (const double*)delta, ldd, col0, step, ignore_mask, (double*)out, N, dof, B));
  return check_launch("thx_vec_retract");
}

int thx_lm_accept_diag(const void* delta, const void* g, const void* diag, int64_t ldv, int32_t n, int32_t B, void* damping,
                       const void* prev_err, const void* new_err, int ellipsoidal, double accept, double down_ratio,
                       double up_ratio, uint8_t* reject, int dtype, void* stream) {
  if (!delta || !g || !damping || !prev_err || !new_err || !reject || (ellipsoidal && !diag))
    return fail("thx_lm_accept_diag: null pointer");
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(lm_accept_diag_kernel<float>, dim3(B), dim3(256), 0, as_stream(stream), (const float*)delta,
                                  (const float*)g, (const float*)diag, ldv, n, (float*)damping, (const float*)prev_err,
                                  (const float*)new_err, ellipsoidal, (float)accept, (float)down_ratio, (float)up_ratio, reject),
               hipLaunchKernelGGL(lm_accept_diag_kernel<double>, dim3(B), dim3(256), 0, as_stream(stream),
                                  (const double*)delta, (const double*)g, (const double*)diag, ldv, n, (double*)damping,
                                  (const double*)prev_err, (const double*)new_err, ellipsoidal, accept, down_ratio, up_ratio,
                                  reject));
  return check_launch("thx_lm_accept_diag");
}

}  // extern "C"
