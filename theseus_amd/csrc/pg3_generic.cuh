// Pose-graph kernels for the SMALL Lie groups, generic over a group functor G: SE2 (2-D SLAM, lie_se2.cuh, records
// [x, y, cos, sin], 3 degrees of freedom), SO3 (rotation-only graphs, lie_so3.cuh, 3x3 records, 3 dof) and SO2 (lie_so2.cuh,
// records [cos, sin], 1 dof).  Same mapping as the SE3 kernels of pg_kernels.hip -- one lane per (pose | edge, problem), batch
// index fastest, owner-computes assembly, no atomics -- with DOF x DOF blocks.  All Lie arithmetic in fp64 registers.
//
// G provides:  DOF (degrees of freedom), REC (scalars per record), X (group element in fp64 registers), Eps (Taylor thresholds),
//              load / store, inv, mul, exp(xi, eps, X, J|nullptr), log_jlog(X, eps, xi, J, want_jac), adjoint(X, A).
#pragma once
#include "common.cuh"
#include "lie.cuh"
#include "robust.cuh"

namespace thx {

// C = A B, N x N row major; the terms of every sum in index order (N = 3: the same expression as lie.cuh's mat3_mul)
template <int N>
__device__ __forceinline__ void matn_mul(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double acc = A[N * i] * B[j];
#pragma unroll
      for (int k = 1; k < N; ++k) acc += A[N * i + k] * B[N * k + j];
      C[N * i + j] = acc;
    }
}

// Between (embodied/measurements/between.py:38-45) with row weights (core/cost_weight.py:125-136):
//   D = v0^-1 v1 ; E = m^-1 D ; e = w . log E ; J1 = w . Jlog(E) ; J0 = -w . Jlog(E) Ad(D^-1)
template <typename G>
__device__ __forceinline__ void between_eval3(const typename G::X& v0, const typename G::X& v1, const typename G::X& meas,
                                              const double* w, const typename G::Eps& eps, double* e, double* J0, double* J1,
                                              bool want_jac) {
  typename G::X v0i, D, mi, E;
  G::inv(v0, v0i);
  G::mul(v0i, v1, D);
  G::inv(meas, mi);
  G::mul(mi, D, E);
  constexpr int N = G::DOF;
  double xi[N], Jl[N * N];
  G::log_jlog(E, eps, xi, Jl, want_jac);
#pragma unroll
  for (int i = 0; i < N; ++i) e[i] = xi[i] * w[i];
  if (!want_jac) return;
  typename G::X Di;
  G::inv(D, Di);
  double Ad[N * N], JA[N * N];
  G::adjoint(Di, Ad);
  matn_mul<N>(Jl, Ad, JA);
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      J1[N * i + j] = Jl[N * i + j] * w[i];
      J0[N * i + j] = -JA[N * i + j] * w[i];
    }
}
// Local / Difference (embodied/misc/local_cost_fn.py:42-61): e = w . log(target^-1 var), J = w . Jlog
template <typename G>
__device__ __forceinline__ void local_eval3(const typename G::X& target, const typename G::X& var, const double* w,
                                            const typename G::Eps& eps, double* e, double* J, bool want_jac) {
  typename G::X ti, D;
  G::inv(target, ti);
  G::mul(ti, var, D);
  constexpr int N = G::DOF;
  double xi[N], Jl[N * N];
  G::log_jlog(D, eps, xi, Jl, want_jac);
#pragma unroll
  for (int i = 0; i < N; ++i) e[i] = xi[i] * w[i];
  if (!want_jac) return;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) J[N * i + j] = Jl[N * i + j] * w[i];
}

template <int N>
__device__ __forceinline__ void m3_tmul_acc(const double* P, const double* Q, double* B) {  // B += P^T Q
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double acc = P[i] * Q[j];
#pragma unroll
      for (int k = 1; k < N; ++k) acc += P[N * k + i] * Q[N * k + j];
      B[N * i + j] += acc;
    }
}
template <int N>
__device__ __forceinline__ void m3_tvec_sub(const double* P, const double* e, double* g) {  // g -= P^T e
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double acc = P[i] * e[0];
#pragma unroll
    for (int k = 1; k < N; ++k) acc += P[N * k + i] * e[k];
    g[i] -= acc;
  }
}
template <typename T, int N>
__device__ __forceinline__ void robustify2(int code, const void* lr, int64_t lr_bs, int64_t entity, int b, int B,
                                           double* ev, double* J0, double* J1) {
  if (code == THX_LOSS_NONE) return;
  double f[N];
  robust_row_scale<N>(code, ev, load_log_radius<T>(lr, entity, b, B, lr_bs), f);
#pragma unroll
  for (int r = 0; r < N; ++r) {
#pragma unroll
    for (int c = 0; c < N; ++c) {
      if (J0) J0[N * r + c] *= f[r];
      if (J1) J1[N * r + c] *= f[r];
    }
    ev[r] *= f[r];
  }
}
template <int N, typename T>
__device__ __forceinline__ void load3(const T* __restrict__ p, double* w) {
#pragma unroll
  for (int i = 0; i < N; ++i) w[i] = (double)p[i];
}

template <typename G, typename T>
__global__ void __launch_bounds__(64)
pg3_assemble_kernel(thx_pg_structure s, thx_pg_data d, T* __restrict__ H, int64_t ld, T* __restrict__ g, typename G::Eps eps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int p = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const T* poses = static_cast<const T*>(d.poses);
  const T* meas = static_cast<const T*>(d.meas);
  const T* wb = static_cast<const T*>(d.w_between);
  const typename G::X Xp = G::load(poses + ((int64_t)p * B + b) * G::REC);
  constexpr int N = G::DOF, NN = N * N;
  double Dg[NN], Off[NN], gv[N];
#pragma unroll
  for (int i = 0; i < NN; ++i) { Dg[i] = 0.0; Off[i] = 0.0; }
#pragma unroll
  for (int i = 0; i < N; ++i) gv[i] = 0.0;
  T* Hb = H + (int64_t)b * ld * ld;
  auto flush = [&](int q) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) Hb[(int64_t)(N * p + r) * ld + N * q + c] = (T)Off[N * r + c];
  };
  const int64_t mB = d.meas_bstride ? B : 1, wB = d.w_between_bstride ? B : 1;
  int cur_q = -1;
  for (int k = s.inc_ptr[p]; k < s.inc_ptr[p + 1]; ++k) {
    const int e = s.inc_edge[k], side = s.inc_side[k], q = s.inc_other[k];
    const typename G::X Xq = G::load(poses + ((int64_t)q * B + b) * G::REC);
    const typename G::X M = G::load(meas + ((int64_t)e * mB) * G::REC + (int64_t)b * d.meas_bstride);
    double w[N], ev[N], J0[NN], J1[NN];
    load3<N>(wb + ((int64_t)e * wB) * N + (int64_t)b * d.w_between_bstride, w);
    const bool lower = q < p;
    if (lower && q != cur_q) {
      if (cur_q >= 0) flush(cur_q);
#pragma unroll
      for (int i = 0; i < NN; ++i) Off[i] = 0.0;
      cur_q = q;
    }
    if (side == 0) {
      between_eval3<G>(Xp, Xq, M, w, eps, ev, J0, J1, true);
      robustify2<T, N>(loss_code(d.robust_between, d.loss_between, e), d.log_radius_between, d.log_radius_between_bstride, e, b, B, ev, J0, J1);
      m3_tmul_acc<N>(J0, J0, Dg);
      m3_tvec_sub<N>(J0, ev, gv);
      if (lower) m3_tmul_acc<N>(J0, J1, Off);
    } else {
      between_eval3<G>(Xq, Xp, M, w, eps, ev, J0, J1, true);
      robustify2<T, N>(loss_code(d.robust_between, d.loss_between, e), d.log_radius_between, d.log_radius_between_bstride, e, b, B, ev, J0, J1);
      m3_tmul_acc<N>(J1, J1, Dg);
      m3_tvec_sub<N>(J1, ev, gv);
      if (lower) m3_tmul_acc<N>(J1, J0, Off);
    }
  }
  if (cur_q >= 0) flush(cur_q);
  const T* tgt = static_cast<const T*>(d.prior_target);
  const T* wp = static_cast<const T*>(d.w_prior);
  const int64_t tB = d.prior_target_bstride ? B : 1, wpB = d.w_prior_bstride ? B : 1;
  for (int k = s.pri_ptr[p]; k < s.pri_ptr[p + 1]; ++k) {
    const int id = s.pri_id[k];
    const typename G::X Tg = G::load(tgt + ((int64_t)id * tB) * G::REC + (int64_t)b * d.prior_target_bstride);
    double w[N], ev[N], J[NN];
    load3<N>(wp + ((int64_t)id * wpB) * N + (int64_t)b * d.w_prior_bstride, w);
    local_eval3<G>(Tg, Xp, w, eps, ev, J, true);
    robustify2<T, N>(loss_code(d.robust_prior, d.loss_prior, id), d.log_radius_prior, d.log_radius_prior_bstride, id, b, B, ev, J, nullptr);
    m3_tmul_acc<N>(J, J, Dg);
    m3_tvec_sub<N>(J, ev, gv);
  }
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int c = 0; c < N; ++c) Hb[(int64_t)(N * p + r) * ld + N * p + c] = (T)Dg[N * r + c];
  T* gb = g + (int64_t)b * (N * s.num_poses) + N * p;
#pragma unroll
  for (int i = 0; i < N; ++i) gb[i] = (T)gv[i];
}

template <typename G, typename T>
__global__ void __launch_bounds__(64)
pg3_error_partial_kernel(thx_pg_structure s, thx_pg_data d, T* __restrict__ partials, typename G::Eps eps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int ch = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const T* poses = static_cast<const T*>(d.poses);
  constexpr int N = G::DOF;
  double acc = 0.0;
  const int E = s.num_edges, K = s.num_priors;
  const int ec = (E + THX_ERR_CHUNKS - 1) / THX_ERR_CHUNKS, e1 = min(E, (ch + 1) * ec);
  const int64_t mB = d.meas_bstride ? B : 1, wB = d.w_between_bstride ? B : 1;
  for (int e = ch * ec; e < e1; ++e) {
    const typename G::X Xi = G::load(poses + ((int64_t)s.edge_i[e] * B + b) * G::REC);
    const typename G::X Xj = G::load(poses + ((int64_t)s.edge_j[e] * B + b) * G::REC);
    const typename G::X M = G::load(static_cast<const T*>(d.meas) + ((int64_t)e * mB) * G::REC + (int64_t)b * d.meas_bstride);
    double w[N], ev[N];
    load3<N>(static_cast<const T*>(d.w_between) + ((int64_t)e * wB) * N + (int64_t)b * d.w_between_bstride, w);
    between_eval3<G>(Xi, Xj, M, w, eps, ev, nullptr, nullptr, false);
    const int code = loss_code(d.robust_between, d.loss_between, e);
    acc += robust_sq_error<N>(code, ev, code ? load_log_radius<T>(d.log_radius_between, e, b, B, d.log_radius_between_bstride) : 0.0);
  }
  const int kc = (K + THX_ERR_CHUNKS - 1) / THX_ERR_CHUNKS, k1 = min(K, (ch + 1) * kc);
  const int64_t tB = d.prior_target_bstride ? B : 1, wpB = d.w_prior_bstride ? B : 1;
  for (int k = ch * kc; k < k1; ++k) {
    const typename G::X X = G::load(poses + ((int64_t)s.prior_pose[k] * B + b) * G::REC);
    const typename G::X Tg = G::load(static_cast<const T*>(d.prior_target) + ((int64_t)k * tB) * G::REC + (int64_t)b * d.prior_target_bstride);
    double w[N], ev[N];
    load3<N>(static_cast<const T*>(d.w_prior) + ((int64_t)k * wpB) * N + (int64_t)b * d.w_prior_bstride, w);
    local_eval3<G>(Tg, X, w, eps, ev, nullptr, false);
    const int code = loss_code(d.robust_prior, d.loss_prior, k);
    acc += robust_sq_error<N>(code, ev, code ? load_log_radius<T>(d.log_radius_prior, k, b, B, d.log_radius_prior_bstride) : 0.0);
  }
  partials[(int64_t)ch * B + b] = (T)acc;
}

template <typename T>
__global__ void pg3_error_reduce_kernel(const T* __restrict__ partials, T* __restrict__ err, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  T acc = T(0);
#pragma unroll 8
  for (int c = 0; c < THX_ERR_CHUNKS; ++c) acc += partials[(int64_t)c * B + b];
  err[b] = T(0.5) * acc;
}

template <typename G, typename T>
__global__ void __launch_bounds__(64)
pg3_jacobians_kernel(thx_pg_structure s, thx_pg_data d, T* __restrict__ J0o, T* __restrict__ J1o, T* __restrict__ ebo,
                     T* __restrict__ Jpo, T* __restrict__ epo, typename G::Eps eps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int c = blockIdx.y;
  const int B = d.batch;
  if (b >= B) return;
  const T* poses = static_cast<const T*>(d.poses);
  constexpr int N = G::DOF, NN = N * N;
  double ev[N], J0[NN], J1[NN], w[N];
  if (c < s.num_edges) {
    const int e = c;
    const int64_t mB = d.meas_bstride ? B : 1, wB = d.w_between_bstride ? B : 1;
    const typename G::X Xi = G::load(poses + ((int64_t)s.edge_i[e] * B + b) * G::REC);
    const typename G::X Xj = G::load(poses + ((int64_t)s.edge_j[e] * B + b) * G::REC);
    const typename G::X M = G::load(static_cast<const T*>(d.meas) + ((int64_t)e * mB) * G::REC + (int64_t)b * d.meas_bstride);
    load3<N>(static_cast<const T*>(d.w_between) + ((int64_t)e * wB) * N + (int64_t)b * d.w_between_bstride, w);
    between_eval3<G>(Xi, Xj, M, w, eps, ev, J0, J1, true);
    robustify2<T, N>(loss_code(d.robust_between, d.loss_between, e), d.log_radius_between, d.log_radius_between_bstride, e, b, B, ev, J0, J1);
    const int64_t o = (int64_t)e * B + b;
#pragma unroll
    for (int k = 0; k < NN; ++k) {
      if (J0o) J0o[o * NN + k] = (T)J0[k];
      if (J1o) J1o[o * NN + k] = (T)J1[k];
    }
    if (ebo) {
#pragma unroll
      for (int r = 0; r < N; ++r) ebo[o * N + r] = (T)ev[r];
    }
  } else {
    const int k = c - s.num_edges;
    const int64_t tB = d.prior_target_bstride ? B : 1, wB = d.w_prior_bstride ? B : 1;
    const typename G::X X = G::load(poses + ((int64_t)s.prior_pose[k] * B + b) * G::REC);
    const typename G::X Tg = G::load(static_cast<const T*>(d.prior_target) + ((int64_t)k * tB) * G::REC + (int64_t)b * d.prior_target_bstride);
    load3<N>(static_cast<const T*>(d.w_prior) + ((int64_t)k * wB) * N + (int64_t)b * d.w_prior_bstride, w);
    local_eval3<G>(Tg, X, w, eps, ev, J0, true);
    robustify2<T, N>(loss_code(d.robust_prior, d.loss_prior, k), d.log_radius_prior, d.log_radius_prior_bstride, k, b, B, ev, J0, nullptr);
    const int64_t o = (int64_t)k * B + b;
#pragma unroll
    for (int q = 0; q < NN; ++q)
      if (Jpo) Jpo[o * NN + q] = (T)J0[q];
    if (epo) {
#pragma unroll
      for (int r = 0; r < N; ++r) epo[o * N + r] = (T)ev[r];
    }
  }
}

template <typename G, typename T>
__global__ void __launch_bounds__(64)
g3_retract_kernel(const T* __restrict__ poses, const T* __restrict__ delta, int64_t ldd, T step,
                   const uint8_t* __restrict__ ignore, T* __restrict__ out, int P, int B, typename G::Eps eps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int p = blockIdx.y;
  if (b >= B) return;
  const T* src = poses + ((int64_t)p * B + b) * G::REC;
  T* dst = out + ((int64_t)p * B + b) * G::REC;
  if (ignore && ignore[b]) {
#pragma unroll
    for (int k = 0; k < G::REC; ++k) dst[k] = src[k];
    return;
  }
  constexpr int N = G::DOF;
  double xi[N];
#pragma unroll
  for (int i = 0; i < N; ++i) xi[i] = (double)(delta[(int64_t)b * ldd + N * p + i] * step);
  typename G::X Ex, Y;
  G::exp(xi, eps, Ex, nullptr);
  G::mul(G::load(src), Ex, Y);
  G::store(dst, Y);
}

// elementwise ops: op 0 exp (xi -> X [, J]), 1 log (X -> xi [, J]), 2 compose, 3 inverse, 4 adjoint
template <typename G, typename T>
__global__ void g3_elementwise_kernel(int op, const T* __restrict__ a, const T* __restrict__ bb, T* __restrict__ o,
                                       T* __restrict__ jac, int64_t count, typename G::Eps eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  constexpr int N = G::DOF, NN = N * N;
  double J[NN];
  if (op == 0) {
    double xi[N];
#pragma unroll
    for (int r = 0; r < N; ++r) xi[r] = (double)a[i * N + r];
    typename G::X X;
    G::exp(xi, eps, X, jac ? J : nullptr);
    G::store(o + i * G::REC, X);
  } else if (op == 1) {
    double xi[N];
    G::log_jlog(G::load(a + i * G::REC), eps, xi, J, jac != nullptr);
#pragma unroll
    for (int r = 0; r < N; ++r) o[i * N + r] = (T)xi[r];
  } else if (op == 2) {
    typename G::X Z;
    G::mul(G::load(a + i * G::REC), G::load(bb + i * G::REC), Z);
    G::store(o + i * G::REC, Z);
  } else if (op == 3) {
    typename G::X Z;
    G::inv(G::load(a + i * G::REC), Z);
    G::store(o + i * G::REC, Z);
  } else {
    G::adjoint(G::load(a + i * G::REC), J);
#pragma unroll
    for (int k = 0; k < NN; ++k) o[i * NN + k] = (T)J[k];
  }
  if (jac && op <= 1) {
#pragma unroll
    for (int k = 0; k < NN; ++k) jac[i * NN + k] = (T)J[k];
  }
}


// ---- host-side launchers shared by the extern "C" entry points of the instantiating files ----------------------------------
template <typename G>
static int check_pg3(const thx_pg_structure* s, const thx_pg_data* d, const char* group) {
  if (!s || !d) return fail("null structure/data");
  if (s->num_poses <= 0 || d->batch <= 0) return fail("empty problem");
  if ((d->meas_bstride != 0 && d->meas_bstride != G::REC) || (d->prior_target_bstride != 0 && d->prior_target_bstride != G::REC))
    return fail("meas / prior_target batch stride must be 0 or the record size of the group");
  if ((d->w_between_bstride != 0 && d->w_between_bstride != G::DOF) || (d->w_prior_bstride != 0 && d->w_prior_bstride != G::DOF))
    return fail("weight batch stride must be 0 or the group's degrees of freedom");
  if (const char* why = check_robust(d)) return fail(why);
  (void)group;
  return 0;
}

template <typename G>
static int pg3_assemble(const thx_pg_structure* s, const thx_pg_data* d, void* H, int64_t ld, void* g, int dtype,
                        typename G::Eps eps, void* stream, const char* name) {
  if (int r = check_pg3<G>(s, d, name)) return r;
  if (!H || !g) return fail("null output");
  if (ld < G::DOF * (int64_t)s->num_poses) return fail("ld < n");
  dim3 grid((d->batch + 63) / 64, s->num_poses), block(64);
  THX_DISPATCH(dtype,
               { hipLaunchKernelGGL((pg3_assemble_kernel<G, float>), grid, block, 0, as_stream(stream), *s, *d, (float*)H, ld,
                                   (float*)g, eps); },
               { hipLaunchKernelGGL((pg3_assemble_kernel<G, double>), grid, block, 0, as_stream(stream), *s, *d, (double*)H,
                                   ld, (double*)g, eps); });
  return check_launch(name);
}

template <typename G>
static int pg3_error(const thx_pg_structure* s, const thx_pg_data* d, void* partials, void* err, int dtype,
                     typename G::Eps eps, void* stream, const char* name) {
  if (int r = check_pg3<G>(s, d, name)) return r;
  if (!partials || !err) return fail("null output");
  dim3 grid((d->batch + 63) / 64, THX_ERR_CHUNKS), block(64);
  const int B = d->batch;
  THX_DISPATCH(dtype,
               {
                 hipLaunchKernelGGL((pg3_error_partial_kernel<G, float>), grid, block, 0, as_stream(stream), *s, *d,
                                    (float*)partials, eps);
                 hipLaunchKernelGGL(pg3_error_reduce_kernel<float>, dim3((B + 255) / 256), dim3(256), 0,
                                    as_stream(stream), (const float*)partials, (float*)err, B);
               },
               {
                 hipLaunchKernelGGL((pg3_error_partial_kernel<G, double>), grid, block, 0, as_stream(stream), *s, *d,
                                    (double*)partials, eps);
                 hipLaunchKernelGGL(pg3_error_reduce_kernel<double>, dim3((B + 255) / 256), dim3(256), 0,
                                    as_stream(stream), (const double*)partials, (double*)err, B);
               });
  return check_launch(name);
}

template <typename G>
static int pg3_jacobians(const thx_pg_structure* s, const thx_pg_data* d, void* J0, void* J1, void* eb, void* Jp, void* ep,
                         int dtype, typename G::Eps eps, void* stream, const char* name) {
  if (int r = check_pg3<G>(s, d, name)) return r;
  dim3 grid((d->batch + 63) / 64, s->num_edges + s->num_priors), block(64);
  if (grid.y == 0) return 0;
  THX_DISPATCH(dtype,
               { hipLaunchKernelGGL((pg3_jacobians_kernel<G, float>), grid, block, 0, as_stream(stream), *s, *d, (float*)J0,
                                   (float*)J1, (float*)eb, (float*)Jp, (float*)ep, eps); },
               { hipLaunchKernelGGL((pg3_jacobians_kernel<G, double>), grid, block, 0, as_stream(stream), *s, *d, (double*)J0,
                                   (double*)J1, (double*)eb, (double*)Jp, (double*)ep, eps); });
  return check_launch(name);
}

template <typename G>
static int g3_retract(const void* poses, const void* delta, int64_t ldd, double step, const uint8_t* ignore_mask, void* out,
                      int32_t P, int32_t B, int dtype, typename G::Eps eps, void* stream, const char* name) {
  if (!poses || !delta || !out || P <= 0 || B <= 0) return fail("bad retract args");
  dim3 grid((B + 63) / 64, P), block(64);
  THX_DISPATCH(dtype,
               { hipLaunchKernelGGL((g3_retract_kernel<G, float>), grid, block, 0, as_stream(stream), (const float*)poses,
                                   (const float*)delta, ldd, (float)step, ignore_mask, (float*)out, P, B, eps); },
               { hipLaunchKernelGGL((g3_retract_kernel<G, double>), grid, block, 0, as_stream(stream), (const double*)poses,
                                   (const double*)delta, ldd, step, ignore_mask, (double*)out, P, B, eps); });
  return check_launch(name);
}

template <typename G>
static int g3_op(int op, const void* a, const void* b, void* out, void* jac, int64_t N, int dtype, typename G::Eps eps,
                 void* stream, const char* name) {
  if (N <= 0) return 0;
  if (op < 0 || op > 4 || !a || !out || (op == 2 && !b)) return fail("elementwise group op: bad arguments");
  dim3 grid((unsigned)((N + 255) / 256)), block(256);
  THX_DISPATCH(dtype,
               { hipLaunchKernelGGL((g3_elementwise_kernel<G, float>), grid, block, 0, as_stream(stream), op, (const float*)a,
                                   (const float*)b, (float*)out, (float*)jac, N, eps); },
               { hipLaunchKernelGGL((g3_elementwise_kernel<G, double>), grid, block, 0, as_stream(stream), op,
                                   (const double*)a, (const double*)b, (double*)out, (double*)jac, N, eps); });
  return check_launch(name);
}

}  // namespace thx
