// Pose-graph kernels for gfx950: fused Between/Local residual + Jacobian + weight + block J^T J / J^T r
// assembly, error metric, Jacobian dump, masked SE3 retraction and the elementwise SE3 ops.
//
// Mapping: one lane per (entity, batch item) with the batch index fastest across the wavefront, so
// every wave works on ONE pose / edge for 64 consecutive problems: structure lookups are
// wave-uniform (scalar loads, no divergence) and pose / measurement reads are 48-byte (f32) records
// at unit stride across lanes -> fully used 128-B lines.  Assembly is "owner computes": the lane
// owning (pose p, problem b) walks p's incident edges (CSR sorted by the other endpoint), rebuilds
// each edge's Jacobians in registers and accumulates its diagonal block, its g segment and the
// off-diagonal blocks H[p][q] for q < p.  No atomics -> bit-reproducible; the Lie arithmetic is
// evaluated twice per edge (once per endpoint), which is ~3 MFLOP per problem against an HBM-bound
// store of the blocks.
#include "common.cuh"
#include "robust.cuh"

namespace thx {
__device__ __forceinline__ void sjac_scale_rows(SJac<double>& J, const double* f) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) { J.a[3 * r + c] *= f[r]; J.c[3 * r + c] *= f[r]; J.d[3 * r + c] *= f[3 + r]; }
}
// robust rescale of a Between / Local evaluation (robust_cost_function.py:115-135); ``code``: robust.cuh
template <typename T>
__device__ __forceinline__ void robustify(int code, const void* lr, int64_t lr_bs, int64_t entity, int b, int B,
                                          double* ev, SJac<double>* J0, SJac<double>* J1) {
  if (code == THX_LOSS_NONE) return;
  double f[6];
  robust_row_scale<6>(code, ev, load_log_radius<T>(lr, entity, b, B, lr_bs), f);
  if (J0) sjac_scale_rows(*J0, f);
  if (J1) sjac_scale_rows(*J1, f);
#pragma unroll
  for (int i = 0; i < 6; ++i) ev[i] *= f[i];
}
// Robust costs in the assembly kernel: the rescale factors f_r = sqrt(rho'(x_r) + eps) only need the residual, so they are
// computed from a residual-only evaluation FIRST and folded into the weights (J' = diag(f) J, e' = diag(f) e  <=>  w' = f w)
// -- the Jacobian pass then runs exactly as for plain costs and nothing but six weights is live across the exp() of the loss
// (rescaling the finished fp64 Jacobians spilled 440 bytes per lane).
template <typename T>
__device__ __forceinline__ void robust_weights(int code, const void* lr, int64_t lr_bs, int64_t entity, int b, int B,
                                               const double* ev, T* w) {
  double f[6];
  robust_row_scale<6>(code, ev, load_log_radius<T>(lr, entity, b, B, lr_bs), f);
#pragma unroll
  for (int i = 0; i < 6; ++i) w[i] = (T)((double)w[i] * f[i]);
}
}  // namespace thx

namespace thx {

std::string& last_error() {
  static thread_local std::string e;
  return e;
}

// 16-byte vector loads of an SE3 record (3x4 row major).
__device__ __forceinline__ void se3_load_vec(const float* __restrict__ p, SE3<float>& X) {
  const float4* q = reinterpret_cast<const float4*>(p);
  float4 r0 = q[0], r1 = q[1], r2 = q[2];
  X.R[0] = r0.x; X.R[1] = r0.y; X.R[2] = r0.z; X.t[0] = r0.w;
  X.R[3] = r1.x; X.R[4] = r1.y; X.R[5] = r1.z; X.t[1] = r1.w;
  X.R[6] = r2.x; X.R[7] = r2.y; X.R[8] = r2.z; X.t[2] = r2.w;
}
__device__ __forceinline__ void se3_load_vec(const double* __restrict__ p, SE3<double>& X) {
  const double2* q = reinterpret_cast<const double2*>(p);
  double2 a0 = q[0], a1 = q[1], b0 = q[2], b1 = q[3], c0 = q[4], c1 = q[5];
  X.R[0] = a0.x; X.R[1] = a0.y; X.R[2] = a1.x; X.t[0] = a1.y;
  X.R[3] = b0.x; X.R[4] = b0.y; X.R[5] = b1.x; X.t[1] = b1.y;
  X.R[6] = c0.x; X.R[7] = c0.y; X.R[8] = c1.x; X.t[2] = c1.y;
}
__device__ __forceinline__ void se3_store_vec(float* __restrict__ p, const SE3<float>& X) {
  float4* q = reinterpret_cast<float4*>(p);
  q[0] = make_float4(X.R[0], X.R[1], X.R[2], X.t[0]);
  q[1] = make_float4(X.R[3], X.R[4], X.R[5], X.t[1]);
  q[2] = make_float4(X.R[6], X.R[7], X.R[8], X.t[2]);
}
__device__ __forceinline__ void se3_store_vec(double* __restrict__ p, const SE3<double>& X) {
  double2* q = reinterpret_cast<double2*>(p);
  q[0] = make_double2(X.R[0], X.R[1]); q[1] = make_double2(X.R[2], X.t[0]);
  q[2] = make_double2(X.R[3], X.R[4]); q[3] = make_double2(X.R[5], X.t[1]);
  q[4] = make_double2(X.R[6], X.R[7]); q[5] = make_double2(X.R[8], X.t[2]);
}
template <typename T>
__device__ __forceinline__ void load6(const T* __restrict__ p, T* w) {
#pragma unroll
  for (int i = 0; i < 6; ++i) w[i] = p[i];
}

// ------------------------------------------------------------------------------------------------
// assembly
// ------------------------------------------------------------------------------------------------
// ROBUST = false compiles the rescale away: the plain-cost kernel keeps its registers (with the rescale inlined the
// fp64 Jacobian chain spilled 440 bytes per lane to scratch: 1.47 -> 1.82 ms and 3x the HBM writes, rocprofv3 PMC)
// one 6x6 block -> its 36 consecutive elements of the block list, 16-byte pieces
template <typename T>
__device__ __forceinline__ void store_block36(T* __restrict__ dst, const T* v) {
  if constexpr (sizeof(T) == 4) {
    float4* q = reinterpret_cast<float4*>(dst);
#pragma unroll
    for (int i = 0; i < 9; ++i) q[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  } else {
    double2* q = reinterpret_cast<double2*>(dst);
#pragma unroll
    for (int i = 0; i < 18; ++i) q[i] = make_double2(v[2 * i], v[2 * i + 1]);
  }
}

// COMPACT: H is the block list (B, nblocks, 36) with problem stride ``ld`` elements (include/theseus_hip.h: thx_hblock_layout),
// a lane writes each of its blocks as 144 (288) contiguous bytes; else the dense (B, ld, ld) frame.
template <typename T, bool ROBUST, bool COMPACT>
__global__ void __launch_bounds__(64)
pg_assemble_kernel(thx_pg_structure s, thx_pg_data d, T* __restrict__ H, int64_t ld, T* __restrict__ g,
                   Eps<T> eps, const int32_t* __restrict__ diag_blk, const int32_t* __restrict__ inc_blk) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int p = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const T* poses = static_cast<const T*>(d.poses);
  const T* meas = static_cast<const T*>(d.meas);
  const T* wb = static_cast<const T*>(d.w_between);
  SE3<T> Xp;
  se3_load_vec(poses + ((int64_t)p * B + b) * 12, Xp);

  T Dg[36], Off[36];
  double gv[6];  // g = -J^T e accumulated in fp64 (lie.cuh, "Evaluation precision")
#pragma unroll
  for (int i = 0; i < 36; ++i) { Dg[i] = T(0); Off[i] = T(0); }
#pragma unroll
  for (int i = 0; i < 6; ++i) gv[i] = 0.0;

  T* Hb = H + (int64_t)b * (COMPACT ? ld : ld * ld);
  const int beg = s.inc_ptr[p], end = s.inc_ptr[p + 1];
  int cur_q = -1, cur_blk = -1;
  for (int k = beg; k < end; ++k) {
    const int e = s.inc_edge[k], side = s.inc_side[k], q = s.inc_other[k];
    SE3<T> Xq, M;
    se3_load_vec(poses + ((int64_t)q * B + b) * 12, Xq);
    const int64_t mB = d.meas_bstride ? B : 1, wB = d.w_between_bstride ? B : 1;
    se3_load_vec(meas + ((int64_t)e * mB) * 12 + (int64_t)b * d.meas_bstride, M);
    T w[6];
    double ev[6];
    load6(wb + ((int64_t)e * wB) * 6 + (int64_t)b * d.w_between_bstride, w);
    SJac<double> J0d, J1d;
    const bool lower = q < p;
    if (lower && q != cur_q) {
      if (cur_q >= 0) {
        if constexpr (COMPACT) {
          store_block36(Hb + (int64_t)cur_blk * 36, Off);
        } else {
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) Hb[(int64_t)(6 * p + r) * ld + 6 * cur_q + c] = Off[6 * r + c];
        }
      }
#pragma unroll
      for (int i = 0; i < 36; ++i) Off[i] = T(0);
      cur_q = q;
      if constexpr (COMPACT) cur_blk = inc_blk[k];
    }
    if (side == 0) {  // p is v0: own Jacobian J0, other J1
      if constexpr (ROBUST) {
        if (const int code = loss_code(d.robust_between, d.loss_between, e)) {
          between_eval_hp<T>(Xp, Xq, M, w, eps, ev, nullptr, nullptr, false);
          robust_weights<T>(code, d.log_radius_between, d.log_radius_between_bstride, e, b, B, ev, w);
        }
      }
      between_eval_hp(Xp, Xq, M, w, eps, ev, &J0d, &J1d, true);
      const SJac<T> J0 = narrow<T>(J0d);
      sjac_tmul_acc(J0, J0, Dg);
      sjac_tvec_sub(J0d, ev, gv);
      if (lower) sjac_tmul_acc(J0, narrow<T>(J1d), Off);
    } else {  // p is v1
      if constexpr (ROBUST) {
        if (const int code = loss_code(d.robust_between, d.loss_between, e)) {
          between_eval_hp<T>(Xq, Xp, M, w, eps, ev, nullptr, nullptr, false);
          robust_weights<T>(code, d.log_radius_between, d.log_radius_between_bstride, e, b, B, ev, w);
        }
      }
      between_eval_hp(Xq, Xp, M, w, eps, ev, &J0d, &J1d, true);
      const SJac<T> J1 = narrow<T>(J1d);
      sjac_tmul_acc(J1, J1, Dg);
      sjac_tvec_sub(J1d, ev, gv);
      if (lower) sjac_tmul_acc(J1, narrow<T>(J0d), Off);
    }
  }
  if (cur_q >= 0) {
    if constexpr (COMPACT) {
      store_block36(Hb + (int64_t)cur_blk * 36, Off);
    } else {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) Hb[(int64_t)(6 * p + r) * ld + 6 * cur_q + c] = Off[6 * r + c];
    }
  }
  // priors on this pose
  const T* tgt = static_cast<const T*>(d.prior_target);
  const T* wp = static_cast<const T*>(d.w_prior);
  for (int k = s.pri_ptr[p]; k < s.pri_ptr[p + 1]; ++k) {
    const int id = s.pri_id[k];
    const int64_t tB = d.prior_target_bstride ? B : 1, wB = d.w_prior_bstride ? B : 1;
    SE3<T> Tg;
    se3_load_vec(tgt + ((int64_t)id * tB) * 12 + (int64_t)b * d.prior_target_bstride, Tg);
    T w[6];
    double ev[6];
    load6(wp + ((int64_t)id * wB) * 6 + (int64_t)b * d.w_prior_bstride, w);
    SJac<double> Jd;
    if constexpr (ROBUST) {
      if (const int code = loss_code(d.robust_prior, d.loss_prior, id)) {
        local_eval_hp<T>(Tg, Xp, w, eps, ev, nullptr, false);
        robust_weights<T>(code, d.log_radius_prior, d.log_radius_prior_bstride, id, b, B, ev, w);
      }
    }
    local_eval_hp(Tg, Xp, w, eps, ev, &Jd, true);
    const SJac<T> J = narrow<T>(Jd);
    sjac_tmul_acc(J, J, Dg);
    sjac_tvec_sub(Jd, ev, gv);
  }
  if constexpr (COMPACT) {
    store_block36(Hb + (int64_t)diag_blk[p] * 36, Dg);
  } else {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) Hb[(int64_t)(6 * p + r) * ld + 6 * p + c] = Dg[6 * r + c];
  }
  T* gb = g + (int64_t)b * (6 * s.num_poses) + 6 * p;
#pragma unroll
  for (int i = 0; i < 6; ++i) gb[i] = (T)gv[i];
}

// ------------------------------------------------------------------------------------------------
// error metric: partial sums over a fixed chunking of the costs, then a fixed-order reduction
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64)
pg_error_partial_kernel(thx_pg_structure s, thx_pg_data d, T* __restrict__ partials, Eps<T> eps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int ch = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const T* poses = static_cast<const T*>(d.poses);
  const T* meas = static_cast<const T*>(d.meas);
  const T* wb = static_cast<const T*>(d.w_between);
  double acc = 0.0;
  const int E = s.num_edges, K = s.num_priors;
  const int ec = (E + THX_ERR_CHUNKS - 1) / THX_ERR_CHUNKS;
  const int e1 = min(E, (ch + 1) * ec);
  const int64_t mB = d.meas_bstride ? B : 1, wB = d.w_between_bstride ? B : 1;
  for (int e = ch * ec; e < e1; ++e) {
    const int i = s.edge_i[e], j = s.edge_j[e];
    SE3<T> Xi, Xj, M;
    se3_load_vec(poses + ((int64_t)i * B + b) * 12, Xi);
    se3_load_vec(poses + ((int64_t)j * B + b) * 12, Xj);
    se3_load_vec(meas + ((int64_t)e * mB) * 12 + (int64_t)b * d.meas_bstride, M);
    T w[6];
    double ev[6];
    load6(wb + ((int64_t)e * wB) * 6 + (int64_t)b * d.w_between_bstride, w);
    between_eval_hp<T>(Xi, Xj, M, w, eps, ev, nullptr, nullptr, false);
    const int code = loss_code(d.robust_between, d.loss_between, e);
    acc += robust_sq_error<6>(code, ev, code ? load_log_radius<T>(d.log_radius_between, e, b, B, d.log_radius_between_bstride) : 0.0);
  }
  const T* tgt = static_cast<const T*>(d.prior_target);
  const T* wp = static_cast<const T*>(d.w_prior);
  const int kc = (K + THX_ERR_CHUNKS - 1) / THX_ERR_CHUNKS;
  const int k1 = min(K, (ch + 1) * kc);
  const int64_t tB = d.prior_target_bstride ? B : 1, wpB = d.w_prior_bstride ? B : 1;
  for (int k = ch * kc; k < k1; ++k) {
    const int p = s.prior_pose[k];
    SE3<T> X, Tg;
    se3_load_vec(poses + ((int64_t)p * B + b) * 12, X);
    se3_load_vec(tgt + ((int64_t)k * tB) * 12 + (int64_t)b * d.prior_target_bstride, Tg);
    T w[6];
    double ev[6];
    load6(wp + ((int64_t)k * wpB) * 6 + (int64_t)b * d.w_prior_bstride, w);
    local_eval_hp<T>(Tg, X, w, eps, ev, nullptr, false);
    const int code = loss_code(d.robust_prior, d.loss_prior, k);
    acc += robust_sq_error<6>(code, ev, code ? load_log_radius<T>(d.log_radius_prior, k, b, B, d.log_radius_prior_bstride) : 0.0);
  }
  partials[(int64_t)ch * B + b] = (T)acc;
}

template <typename T>
__global__ void pg_error_reduce_kernel(const T* __restrict__ partials, T* __restrict__ err, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  T acc = T(0);
#pragma unroll 8
  for (int c = 0; c < THX_ERR_CHUNKS; ++c) acc += partials[(int64_t)c * B + b];
  err[b] = T(0.5) * acc;
}

// ------------------------------------------------------------------------------------------------
// Jacobian / residual dump
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64)
pg_jacobians_kernel(thx_pg_structure s, thx_pg_data d, T* __restrict__ J0o, T* __restrict__ J1o,
                    T* __restrict__ ebo, T* __restrict__ Jpo, T* __restrict__ epo, Eps<T> eps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int c = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const T* poses = static_cast<const T*>(d.poses);
  double ev[6];
  T M36[36];
  if (c < s.num_edges) {
    const int e = c, i = s.edge_i[e], j = s.edge_j[e];
    const int64_t mB = d.meas_bstride ? B : 1, wB = d.w_between_bstride ? B : 1;
    SE3<T> Xi, Xj, M;
    se3_load_vec(poses + ((int64_t)i * B + b) * 12, Xi);
    se3_load_vec(poses + ((int64_t)j * B + b) * 12, Xj);
    se3_load_vec(static_cast<const T*>(d.meas) + ((int64_t)e * mB) * 12 + (int64_t)b * d.meas_bstride, M);
    T w[6];
    load6(static_cast<const T*>(d.w_between) + ((int64_t)e * wB) * 6 + (int64_t)b * d.w_between_bstride, w);
    SJac<double> J0d, J1d;
    between_eval_hp(Xi, Xj, M, w, eps, ev, &J0d, &J1d, true);
    robustify<T>(loss_code(d.robust_between, d.loss_between, e), d.log_radius_between, d.log_radius_between_bstride, e, b, B, ev, &J0d, &J1d);
    const SJac<T> J0 = narrow<T>(J0d), J1 = narrow<T>(J1d);
    const int64_t o = (int64_t)e * B + b;
    if (J0o) {
      sjac_dense(J0, M36);
#pragma unroll
      for (int k = 0; k < 36; ++k) J0o[o * 36 + k] = M36[k];
    }
    if (J1o) {
      sjac_dense(J1, M36);
#pragma unroll
      for (int k = 0; k < 36; ++k) J1o[o * 36 + k] = M36[k];
    }
    if (ebo) {
#pragma unroll
      for (int k = 0; k < 6; ++k) ebo[o * 6 + k] = (T)ev[k];
    }
  } else {
    const int k = c - s.num_edges, p = s.prior_pose[k];
    const int64_t tB = d.prior_target_bstride ? B : 1, wB = d.w_prior_bstride ? B : 1;
    SE3<T> X, Tg;
    se3_load_vec(poses + ((int64_t)p * B + b) * 12, X);
    se3_load_vec(static_cast<const T*>(d.prior_target) + ((int64_t)k * tB) * 12 + (int64_t)b * d.prior_target_bstride, Tg);
    T w[6];
    load6(static_cast<const T*>(d.w_prior) + ((int64_t)k * wB) * 6 + (int64_t)b * d.w_prior_bstride, w);
    SJac<double> Jd;
    local_eval_hp(Tg, X, w, eps, ev, &Jd, true);
    robustify<T>(loss_code(d.robust_prior, d.loss_prior, k), d.log_radius_prior, d.log_radius_prior_bstride, k, b, B, ev, &Jd, nullptr);
    const SJac<T> J = narrow<T>(Jd);
    const int64_t o = (int64_t)k * B + b;
    if (Jpo) {
      sjac_dense(J, M36);
#pragma unroll
      for (int q = 0; q < 36; ++q) Jpo[o * 36 + q] = M36[q];
    }
    if (epo) {
#pragma unroll
      for (int q = 0; q < 6; ++q) epo[o * 6 + q] = (T)ev[q];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// retraction
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64)
se3_retract_kernel(const T* __restrict__ poses, const T* __restrict__ delta, int64_t ldd, T step,
                   const uint8_t* __restrict__ ignore, T* __restrict__ out, int P, int B, Eps<T> eps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B) return;
  SE3<T> X, Y;
  se3_load_vec(poses + ((int64_t)p * B + b) * 12, X);
  if (ignore && ignore[b]) {
    se3_store_vec(out + ((int64_t)p * B + b) * 12, X);
    return;
  }
  // the scaled step delta * step_size is formed in T like the reference (nonlinear_least_squares.py:167-175)
  double xi[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) xi[i] = (double)(delta[(int64_t)b * ldd + 6 * p + i] * step);
  ExpCoef<double> c;
  double Ct;
  SE3<double> Exd, Yd;
  se3_exp(xi, widen(eps), Exd, c, Ct);
  se3_mul(widen(X), Exd, Yd);
  Y = narrow<T>(Yd);
  se3_store_vec(out + ((int64_t)p * B + b) * 12, Y);
}

// ------------------------------------------------------------------------------------------------
// elementwise SE3 ops
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void se3_exp_kernel(const T* __restrict__ xi, T* __restrict__ X, T* __restrict__ jac, int64_t N,
                               Eps<T> eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  T vt[6];
  load6(xi + i * 6, vt);
  double v[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) v[k] = (double)vt[k];
  SE3<double> E;
  ExpCoef<double> c;
  double Ct;
  se3_exp(v, widen(eps), E, c, Ct);
  se3_store_vec(X + i * 12, narrow<T>(E));
  if (jac) {
    double J[36];
    se3_jexp(v, E, c, Ct, J);
#pragma unroll
    for (int k = 0; k < 36; ++k) jac[i * 36 + k] = (T)J[k];
  }
}

template <typename T>
__global__ void se3_log_kernel(const T* __restrict__ X, T* __restrict__ xi, T* __restrict__ jac, int64_t N,
                               Eps<T> eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  SE3<T> G;
  se3_load_vec(X + i * 12, G);
  double v[6], Jr[9], Jt[9];
  se3_log_jlog(widen(G), widen(eps), v, Jr, Jt, jac != nullptr);
#pragma unroll
  for (int k = 0; k < 6; ++k) xi[i * 6 + k] = (T)v[k];
  if (jac) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        jac[i * 36 + 6 * r + c] = (T)Jr[3 * r + c];
        jac[i * 36 + 6 * r + 3 + c] = (T)Jt[3 * r + c];
        jac[i * 36 + 6 * (r + 3) + c] = T(0);
        jac[i * 36 + 6 * (r + 3) + 3 + c] = (T)Jr[3 * r + c];
      }
  }
}

template <typename T>
__global__ void se3_compose_kernel(const T* __restrict__ X, const T* __restrict__ Y, T* __restrict__ Z, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  SE3<T> a, b, c;
  se3_load_vec(X + i * 12, a);
  se3_load_vec(Y + i * 12, b);
  se3_mul(a, b, c);
  se3_store_vec(Z + i * 12, c);
}

template <typename T>
__global__ void se3_inverse_kernel(const T* __restrict__ X, T* __restrict__ Y, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  SE3<T> a, b;
  se3_load_vec(X + i * 12, a);
  se3_inv(a, b);
  se3_store_vec(Y + i * 12, b);
}

template <typename T>
__global__ void se3_adjoint_kernel(const T* __restrict__ X, T* __restrict__ A, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  SE3<T> a;
  se3_load_vec(X + i * 12, a);
  T Hm[9] = {T(0), -a.t[2], a.t[1], a.t[2], T(0), -a.t[0], -a.t[1], a.t[0], T(0)};
  T HR[9];
  mat3_mul(Hm, a.R, HR);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      A[i * 36 + 6 * r + c] = a.R[3 * r + c];
      A[i * 36 + 6 * r + 3 + c] = HR[3 * r + c];
      A[i * 36 + 6 * (r + 3) + c] = T(0);
      A[i * 36 + 6 * (r + 3) + 3 + c] = a.R[3 * r + c];
    }
}

static int check_pg(const thx_pg_structure* s, const thx_pg_data* d) {
  if (!s || !d) return fail("null structure/data");
  if (s->num_poses <= 0 || d->batch <= 0) return fail("empty problem");
  if (d->meas_bstride != 0 && d->meas_bstride != 12) return fail("meas_bstride must be 0 or 12");
  if (d->prior_target_bstride != 0 && d->prior_target_bstride != 12) return fail("prior_target_bstride must be 0 or 12");
  if (d->w_between_bstride != 0 && d->w_between_bstride != 6) return fail("w_between_bstride must be 0 or 6");
  if (d->w_prior_bstride != 0 && d->w_prior_bstride != 6) return fail("w_prior_bstride must be 0 or 6");
  if (const char* why = check_robust(d)) return fail(why);
  return 0;
}

// dst word w of record (k, b) <- src word w where mask[b] (grid-stride over 4-byte words: HBM-bound, coalesced)
__global__ void __launch_bounds__(256)
copy_where_kernel(const uint8_t* __restrict__ mask, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                  int64_t words, int B, int wpr) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256) {
    const int b = (int)((i / wpr) % B);
    if (mask[b]) dst[i] = src[i];
  }
}

// block list -> dense frame: one thread per (problem, piece element); the expansion walks the tile pieces -- every element of
// every piece that lies inside its tile is written exactly once.
template <typename T>
__global__ void __launch_bounds__(256)
hblocks_expand_kernel(thx_hblock_layout lay, const T* __restrict__ Hc, int64_t bstride, T* __restrict__ H, int64_t ld, int npieces) {
  const int b = blockIdx.y;
  const int bb = lay.bd * lay.bd;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= npieces * bb) return;
  const int pc = idx / bb, e = idx % bb;
  // tile of piece pc: binary search in tile_ptr
  int lo = 0, hi = lay.ntiles * (lay.ntiles + 1) / 2;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (lay.tile_ptr[mid] <= pc) lo = mid; else hi = mid;
  }
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= lo) ++ti;
  const int tj = lo - ti * (ti + 1) / 2;
  const int rc = lay.piece_rc[pc];
  const int r = (int)(short)(rc >> 16) + e / lay.bd, c = (int)(short)(rc & 0xffff) + e % lay.bd;
  if (r < 0 || r >= THX_TILE || c < 0 || c >= THX_TILE) return;
  H[(int64_t)b * ld * ld + (int64_t)(ti * THX_TILE + r) * ld + tj * THX_TILE + c] =
      Hc[(int64_t)b * bstride + (int64_t)lay.piece_blk[pc] * bb + e];
}

template <typename T>
__global__ void __launch_bounds__(256)
hblocks_diag_kernel(thx_hblock_layout lay, const T* __restrict__ Hc, int64_t bstride, T* __restrict__ d, int64_t ldv) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= lay.nvars * lay.bd) return;
  const int k = i / lay.bd, r = i % lay.bd;
  d[(int64_t)b * ldv + i] = Hc[(int64_t)b * bstride + (int64_t)lay.diag_blk[k] * lay.bd * lay.bd + r * (lay.bd + 1)];
}

}  // namespace thx

using namespace thx;

extern "C" {

const char* thx_last_error(void) { return last_error().c_str(); }
int thx_abi_version(void) { return 24; }

int thx_copy_where(const uint8_t* mask, const void* src, void* dst, int64_t N, int32_t B, int32_t record_bytes, void* stream) {
  if (!mask || !src || !dst || N < 0 || B <= 0 || record_bytes <= 0 || (record_bytes & 3))
    return fail("thx_copy_where: bad arguments");
  const int64_t words = N * (int64_t)B * (record_bytes / 4);
  if (words == 0) return 0;
  const int64_t blocks = (words + 255) / 256;
  hipLaunchKernelGGL(copy_where_kernel, dim3((unsigned)(blocks < 65536 * 16 ? blocks : 65536 * 16)), dim3(256), 0,
                     as_stream(stream), mask, (const uint32_t*)src, (uint32_t*)dst, words, B, record_bytes / 4);
  return check_launch("thx_copy_where");
}

int thx_pg_assemble(const thx_pg_structure* s, const thx_pg_data* d, void* H, int64_t ld, void* g, int dtype,
                    const thx_lie_eps* eps, void* stream) {
  if (int r = check_pg(s, d)) return r;
  if (!H || !g || !eps) return fail("null output");
  if (ld < 6 * (int64_t)s->num_poses) return fail("ld < n");
  dim3 grid((d->batch + 63) / 64, s->num_poses), block(64);
  const bool robust = d->robust_between != THX_LOSS_NONE || d->robust_prior != THX_LOSS_NONE;
#define THX_ASM(T, R) hipLaunchKernelGGL((pg_assemble_kernel<T, R, false>), grid, block, 0, as_stream(stream), *s, *d, (T*)H, ld, \
                                         (T*)g, make_eps<T>(eps), (const int32_t*)nullptr, (const int32_t*)nullptr)
  THX_DISPATCH(dtype, { if (robust) THX_ASM(float, true); else THX_ASM(float, false); },
               { if (robust) THX_ASM(double, true); else THX_ASM(double, false); });
#undef THX_ASM
  return check_launch("thx_pg_assemble");
}

int thx_pg_assemble_blocks(const thx_pg_structure* s, const thx_pg_data* d, const thx_hblock_layout* layout, void* Hc,
                           int64_t bstride, void* g, int dtype, const thx_lie_eps* eps, void* stream) {
  if (int r = check_pg(s, d)) return r;
  if (!Hc || !g || !eps || !layout || !layout->diag_blk || !layout->inc_blk) return fail("thx_pg_assemble_blocks: null pointer");
  if (layout->bd != 6 || layout->nvars != s->num_poses) return fail("thx_pg_assemble_blocks: the block layout is not this (SE3) graph's");
  if (bstride < 36 * (int64_t)layout->nblocks || (bstride % 4) != 0) return fail("thx_pg_assemble_blocks: bstride");
  dim3 grid((d->batch + 63) / 64, s->num_poses), block(64);
  const bool robust = d->robust_between != THX_LOSS_NONE || d->robust_prior != THX_LOSS_NONE;
#define THX_ASM(T, R) hipLaunchKernelGGL((pg_assemble_kernel<T, R, true>), grid, block, 0, as_stream(stream), *s, *d, (T*)Hc, bstride, \
                                         (T*)g, make_eps<T>(eps), layout->diag_blk, layout->inc_blk)
  THX_DISPATCH(dtype, { if (robust) THX_ASM(float, true); else THX_ASM(float, false); },
               { if (robust) THX_ASM(double, true); else THX_ASM(double, false); });
#undef THX_ASM
  return check_launch("thx_pg_assemble_blocks");
}

static int check_layout(const thx_hblock_layout* l, const char* who) {
  if (!l || !l->diag_blk || !l->tile_ptr || !l->piece_blk || !l->piece_rc || l->nblocks <= 0 || l->bd <= 0 || l->nvars <= 0 ||
      l->ntiles != (l->nvars * l->bd + THX_TILE - 1) / THX_TILE) {
    last_error() = std::string(who) + ": incomplete / inconsistent block layout";
    return -1;
  }
  return 0;
}

int thx_hblocks_expand(const thx_hblock_layout* layout, const void* Hc, int64_t bstride, int32_t B, void* H, int64_t ld,
                       int dtype, void* stream) {
  if (int r = check_layout(layout, "thx_hblocks_expand")) return r;
  if (!Hc || !H || B <= 0 || ld < (int64_t)layout->nvars * layout->bd) return fail("thx_hblocks_expand: bad arguments");
  int npieces = 0;   // (host copy of the last tile_ptr entry: one small synchronous read; this entry point is off the hot path)
  const int ntl = layout->ntiles * (layout->ntiles + 1) / 2;
  if (hipMemcpy(&npieces, layout->tile_ptr + ntl, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
    return fail("thx_hblocks_expand: cannot read the piece count");
  dim3 grid((npieces * layout->bd * layout->bd + 255) / 256, B), block(256);
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(hblocks_expand_kernel<float>, grid, block, 0, as_stream(stream), *layout, (const float*)Hc,
                                  bstride, (float*)H, ld, npieces),
               hipLaunchKernelGGL(hblocks_expand_kernel<double>, grid, block, 0, as_stream(stream), *layout, (const double*)Hc,
                                  bstride, (double*)H, ld, npieces));
  return check_launch("thx_hblocks_expand");
}

int thx_hblocks_diag(const thx_hblock_layout* layout, const void* Hc, int64_t bstride, int32_t B, void* d, int64_t ldv,
                     int dtype, void* stream) {
  if (int r = check_layout(layout, "thx_hblocks_diag")) return r;
  if (!Hc || !d || B <= 0 || ldv < (int64_t)layout->nvars * layout->bd) return fail("thx_hblocks_diag: bad arguments");
  dim3 grid((layout->nvars * layout->bd + 255) / 256, B), block(256);
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(hblocks_diag_kernel<float>, grid, block, 0, as_stream(stream), *layout, (const float*)Hc, bstride,
                                  (float*)d, ldv),
               hipLaunchKernelGGL(hblocks_diag_kernel<double>, grid, block, 0, as_stream(stream), *layout, (const double*)Hc,
                                  bstride, (double*)d, ldv));
  return check_launch("thx_hblocks_diag");
}

int thx_pg_error(const thx_pg_structure* s, const thx_pg_data* d, void* partials, void* err, int dtype,
                 const thx_lie_eps* eps, void* stream) {
  if (int r = check_pg(s, d)) return r;
  if (!partials || !err || !eps) return fail("null output");
  dim3 grid((d->batch + 63) / 64, THX_ERR_CHUNKS), block(64);
  const int B = d->batch;
  THX_DISPATCH(dtype,
               {
                 hipLaunchKernelGGL(pg_error_partial_kernel<float>, grid, block, 0, as_stream(stream), *s, *d,
                                    (float*)partials, make_eps<float>(eps));
                 hipLaunchKernelGGL(pg_error_reduce_kernel<float>, dim3((B + 255) / 256), dim3(256), 0,
                                    as_stream(stream), (const float*)partials, (float*)err, B);
               },
               {
                 hipLaunchKernelGGL(pg_error_partial_kernel<double>, grid, block, 0, as_stream(stream), *s, *d,
                                    (double*)partials, make_eps<double>(eps));
                 hipLaunchKernelGGL(pg_error_reduce_kernel<double>, dim3((B + 255) / 256), dim3(256), 0,
                                    as_stream(stream), (const double*)partials, (double*)err, B);
               });
  return check_launch("thx_pg_error");
}

int thx_pg_jacobians(const thx_pg_structure* s, const thx_pg_data* d, void* J0, void* J1, void* eb, void* Jp,
                     void* ep, int dtype, const thx_lie_eps* eps, void* stream) {
  if (int r = check_pg(s, d)) return r;
  if (!eps) return fail("null eps");
  dim3 grid((d->batch + 63) / 64, s->num_edges + s->num_priors), block(64);
  if (grid.y == 0) return 0;
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(pg_jacobians_kernel<float>, grid, block, 0, as_stream(stream), *s, *d,
                                  (float*)J0, (float*)J1, (float*)eb, (float*)Jp, (float*)ep, make_eps<float>(eps)),
               hipLaunchKernelGGL(pg_jacobians_kernel<double>, grid, block, 0, as_stream(stream), *s, *d,
                                  (double*)J0, (double*)J1, (double*)eb, (double*)Jp, (double*)ep,
                                  make_eps<double>(eps)));
  return check_launch("thx_pg_jacobians");
}

int thx_se3_retract(const void* poses, const void* delta, int64_t ldd, double step, const uint8_t* ignore_mask,
                    void* out, int32_t P, int32_t B, int dtype, const thx_lie_eps* eps, void* stream) {
  if (!poses || !delta || !out || !eps || P <= 0 || B <= 0) return fail("bad retract args");
  dim3 grid((B + 63) / 64, P), block(64);
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(se3_retract_kernel<float>, grid, block, 0, as_stream(stream), (const float*)poses,
                                  (const float*)delta, ldd, (float)step, ignore_mask, (float*)out, P, B,
                                  make_eps<float>(eps)),
               hipLaunchKernelGGL(se3_retract_kernel<double>, grid, block, 0, as_stream(stream),
                                  (const double*)poses, (const double*)delta, ldd, step, ignore_mask, (double*)out,
                                  P, B, make_eps<double>(eps)));
  return check_launch("thx_se3_retract");
}

#define THX_EW_GRID dim3 grid((unsigned)((N + 255) / 256)), block(256)

int thx_se3_exp(const void* xi, void* X, void* jac, int64_t N, int dtype, const thx_lie_eps* eps, void* stream) {
  if (N <= 0) return 0;
  if (!xi || !X || !eps) return fail("null arg");
  THX_EW_GRID;
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(se3_exp_kernel<float>, grid, block, 0, as_stream(stream), (const float*)xi,
                                  (float*)X, (float*)jac, N, make_eps<float>(eps)),
               hipLaunchKernelGGL(se3_exp_kernel<double>, grid, block, 0, as_stream(stream), (const double*)xi,
                                  (double*)X, (double*)jac, N, make_eps<double>(eps)));
  return check_launch("thx_se3_exp");
}

int thx_se3_log(const void* X, void* xi, void* jac, int64_t N, int dtype, const thx_lie_eps* eps, void* stream) {
  if (N <= 0) return 0;
  if (!xi || !X || !eps) return fail("null arg");
  THX_EW_GRID;
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(se3_log_kernel<float>, grid, block, 0, as_stream(stream), (const float*)X,
                                  (float*)xi, (float*)jac, N, make_eps<float>(eps)),
               hipLaunchKernelGGL(se3_log_kernel<double>, grid, block, 0, as_stream(stream), (const double*)X,
                                  (double*)xi, (double*)jac, N, make_eps<double>(eps)));
  return check_launch("thx_se3_log");
}

int thx_se3_compose(const void* X, const void* Y, void* Z, int64_t N, int dtype, void* stream) {
  if (N <= 0) return 0;
  if (!X || !Y || !Z) return fail("null arg");
  THX_EW_GRID;
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(se3_compose_kernel<float>, grid, block, 0, as_stream(stream), (const float*)X,
                                  (const float*)Y, (float*)Z, N),
               hipLaunchKernelGGL(se3_compose_kernel<double>, grid, block, 0, as_stream(stream), (const double*)X,
                                  (const double*)Y, (double*)Z, N));
  return check_launch("thx_se3_compose");
}

int thx_se3_inverse(const void* X, void* Y, int64_t N, int dtype, void* stream) {
  if (N <= 0) return 0;
  if (!X || !Y) return fail("null arg");
  THX_EW_GRID;
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(se3_inverse_kernel<float>, grid, block, 0, as_stream(stream), (const float*)X,
                                  (float*)Y, N),
               hipLaunchKernelGGL(se3_inverse_kernel<double>, grid, block, 0, as_stream(stream), (const double*)X,
                                  (double*)Y, N));
  return check_launch("thx_se3_inverse");
}

int thx_se3_adjoint(const void* X, void* A, int64_t N, int dtype, void* stream) {
  if (N <= 0) return 0;
  if (!X || !A) return fail("null arg");
  THX_EW_GRID;
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(se3_adjoint_kernel<float>, grid, block, 0, as_stream(stream), (const float*)X,
                                  (float*)A, N),
               hipLaunchKernelGGL(se3_adjoint_kernel<double>, grid, block, 0, as_stream(stream), (const double*)X,
                                  (double*)A, N));
  return check_launch("thx_se3_adjoint");
}

}  // extern "C"
