// Batched damped dense Cholesky (factor + solves) for gfx950, one problem per workgroup-chain.
//
// Replaces DenseSolver._apply_damping + CholeskyDenseSolver._solve_sytem
// (theseus/optimizer/linear/dense_solver.py:38-64,159-161) for a batch of B SPD matrices of order n.
//
// Algorithm: tiled LEFT-LOOKING Cholesky with 128x128 tiles, one kernel launch pair per block
// column j (the batch supplies the parallelism: B x (N-j) workgroups per launch, no inter-workgroup
// communication inside a launch):
//   chol_diag(j)    : S = H_jj + damping - L_j,0:j L_j,0:j^T   (MFMA K-loop)
//                     L_jj = chol(S)                            (register-resident, row per lane pair)
//   chol_offdiag(j) : P = H_ij - L_i,0:j L_j,0:j^T              (MFMA K-loop)
//                     L_ij = P L_jj^-T                          (register-resident substitution)
// Left-looking means every tile of L is written exactly once and the trailing matrix is never
// re-read: per problem the K-loops stream  sum_j (N-j) * 2*128*(128 j)  elements, i.e. algorithmic
// intensity T/4 = 32 flop/B in fp32 -- above the 157 TF / 8 TB/s ridge -- so the kernel is MFMA bound.
//
// MFMA mapping (f32: v_mfma_f32_32x32x2_f32, f64: v_mfma_f64_16x16x4_f64): a 256-thread workgroup is
// 4 waves; wave w owns tile rows [32w, 32w+32) x all 128 columns and computes the TRANSPOSED product
// block D = L_j-rows * L_i-rows^T, so that the MFMA result layout puts matrix row r = 32w + (lane&31)
// in lane pair (lane, lane^32) -- lane group g = lane>>5 holds columns c with ((c>>2)&1) == g.
// That "row per lane pair" layout is exactly what the in-register Cholesky / triangular solve need,
// so the tile never round-trips through LDS in fp32.
#include "common.cuh"

#include <utility>

namespace thx {

constexpr int TILE = THX_TILE;

// compile-time loop: the body sees the index as a constant expression, so every register-array
// subscript below is static (a plain `#pragma unroll` over 128 fat iterations is refused by the
// optimiser and would push the row registers to scratch)
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <typename T>
struct CT;
template <>
struct CT<float> {
  static constexpr int KB = 32, LDT = 36, VEC = 4, LDC = 132;
  using V = float4;
};
template <>
struct CT<double> {
  static constexpr int KB = 16, LDT = 18, VEC = 2, LDC = 132;
  using V = double2;
};

// sum of a value over the two lanes (l, l^32) that share a matrix row
__device__ __forceinline__ float pair_sum(float x) {
  unsigned u = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ double pair_sum(double x) {
  unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

__device__ __forceinline__ float t_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ double t_rcp(double x) { return 1.0 / x; }

// canonical register slot -> tile column: idx in [0,64), lane group g in {0,1}
__device__ __forceinline__ constexpr int col_of(int idx, int g) {
  return 32 * (idx >> 4) + 8 * ((idx & 15) >> 2) + 4 * g + (idx & 3);
}

// ------------------------------------------------------------------------------------------------
// K-loop engines.  Stage rows through LDS (register prefetch of the next chunk), accumulate
// acc[r][c] = sum_k Brows[r][k] * Arows[c][k] in the transposed MFMA layout.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct Engine;

template <>
struct Engine<float> {
  struct Acc {
    f32x16 v[4];
  };
  static __device__ __forceinline__ void zero(Acc& a) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) a.v[i][j] = 0.f;
  }
  static __device__ __forceinline__ void chunk(const float* sA, const float* sBw, Acc& acc, int lane) {
    const int rl = lane & 31, g = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float4 fb = *reinterpret_cast<const float4*>(sBw + rl * 36 + 8 * ks + 4 * g);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const float4 fa = *reinterpret_cast<const float4*>(sA + (32 * cb + rl) * 36 + 8 * ks + 4 * g);
        acc.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc.v[cb], 0, 0, 0);
        acc.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc.v[cb], 0, 0, 0);
        acc.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc.v[cb], 0, 0, 0);
        acc.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc.v[cb], 0, 0, 0);
      }
    }
  }
  // MFMA layout == canonical layout: a[16 cb + rho] = acc[cb][rho]
  static __device__ __forceinline__ void canonical(const Acc& acc, float* a, float* /*lds*/, int /*wave*/, int /*lane*/) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[16 * cb + r] = acc.v[cb][r];
  }
  static constexpr size_t canonical_lds_bytes = 0;
};

template <>
struct Engine<double> {
  struct Acc {
    f64x4 v[2][8];
  };
  static __device__ __forceinline__ void zero(Acc& a) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) a.v[h][i][j] = 0.0;
  }
  static __device__ __forceinline__ void chunk(const double* sA, const double* sBw, Acc& acc, int lane) {
    const int rl = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      double2 fb[2];
      fb[0] = *reinterpret_cast<const double2*>(sBw + rl * 18 + 8 * ks + 2 * kq);
      fb[1] = *reinterpret_cast<const double2*>(sBw + (16 + rl) * 18 + 8 * ks + 2 * kq);
#pragma unroll
      for (int cb = 0; cb < 8; ++cb) {
        const double2 fa = *reinterpret_cast<const double2*>(sA + (16 * cb + rl) * 18 + 8 * ks + 2 * kq);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          acc.v[h][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa.x, fb[h].x, acc.v[h][cb], 0, 0, 0);
          acc.v[h][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa.y, fb[h].y, acc.v[h][cb], 0, 0, 0);
        }
      }
    }
  }
  // f64 MFMA layout: D[i = (lane>>4) + 4 rho][j = lane&15]; with D = Arows-block x Brows-block^T this is
  // C[r = 32w + 16h + (lane&15)][c = 16cb + 4 rho + (lane>>4)].  Re-shape to the canonical
  // row-per-lane-pair layout through an LDS tile [128][LDC].
  static __device__ __forceinline__ void canonical(const Acc& acc, double* a, double* lds, int wave, int lane) {
    const int rl = lane & 15, kq = lane >> 4;
    __syncthreads();  // staging buffers (aliased) no longer in use
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho)
          lds[(32 * wave + 16 * h + rl) * 132 + 16 * cb + 4 * rho + kq] = acc.v[h][cb][rho];
    __syncthreads();
    const int r = 32 * wave + (lane & 31), g = lane >> 5;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const double2 x0 = *reinterpret_cast<const double2*>(lds + r * 132 + 8 * m + 4 * g);
      const double2 x1 = *reinterpret_cast<const double2*>(lds + r * 132 + 8 * m + 4 * g + 2);
      const int idx = 16 * (m >> 2) + 4 * (m & 3);
      a[idx] = x0.x; a[idx + 1] = x0.y; a[idx + 2] = x1.x; a[idx + 3] = x1.y;
    }
    __syncthreads();
  }
  static constexpr size_t canonical_lds_bytes = (size_t)128 * 132 * sizeof(double);
};

template <typename T, bool SAME>
__device__ __forceinline__ void kloop(const T* __restrict__ Arows, int validA, const T* __restrict__ Brows,
                                      int validB, int64_t ld, int K, T* sA, T* sB,
                                      typename Engine<T>::Acc& acc, int tid) {
  using C = CT<T>;
  const int lrow = tid >> 3, lc = tid & 7;
  const int wave = tid >> 6, lane = tid & 63;
  uint4 ra[4], rb[4];
  // rows outside the matrix are read from a clamped (in-bounds) row and zeroed by value: selecting
  // between a global pointer and a local zero makes hipcc emit flat loads through scratch
  int64_t offA[4], offB[4];
  bool okA[4], okB[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int row = lrow + 32 * u;
    okA[u] = row < validA;
    okB[u] = row < validB;
    offA[u] = (int64_t)(okA[u] ? row : 0) * ld + lc * C::VEC;
    offB[u] = (int64_t)(okB[u] ? row : 0) * ld + lc * C::VEC;
  }
  auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      uint4 va = *reinterpret_cast<const uint4*>(Arows + offA[u] + k0);
      ra[u] = okA[u] ? va : make_uint4(0, 0, 0, 0);
      if (!SAME) {
        uint4 vb = *reinterpret_cast<const uint4*>(Brows + offB[u] + k0);
        rb[u] = okB[u] ? vb : make_uint4(0, 0, 0, 0);
      }
    }
  };
  const int nk = K / C::KB;
  if (nk > 0) gload(0);
  for (int kc = 0; kc < nk; ++kc) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = lrow + 32 * u;
      *reinterpret_cast<uint4*>(sA + row * C::LDT + lc * C::VEC) = ra[u];
      if (!SAME) *reinterpret_cast<uint4*>(sB + row * C::LDT + lc * C::VEC) = rb[u];
    }
    __syncthreads();
    if (kc + 1 < nk) gload((kc + 1) * C::KB);
    Engine<T>::chunk(sA, (SAME ? sA : sB) + 32 * wave * C::LDT, acc, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// chol_diag: SYRK + in-register Cholesky of the 128x128 diagonal tile of block column j
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256, sizeof(T) == 4 ? 2 : 1)
chol_diag_kernel(const T* __restrict__ H, T* __restrict__ L, T* __restrict__ diagT, const T* __restrict__ damping,
                 int ellipsoidal, T damping_eps, int32_t* __restrict__ info, int n, int64_t ld, int j, int ntiles) {
  using C = CT<T>;
  using V = typename C::V;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t mat = (int64_t)b * ld * ld;
  const int row0 = j * TILE;
  const int valid = min(TILE, n - row0);
  T* sA = smem;  // [128][LDT]; the f64 canonicalisation tile aliases it
  const size_t stage_elems = (size_t)128 * C::LDT;
  const size_t canon_elems = Engine<T>::canonical_lds_bytes / sizeof(T);
  T* colbuf = smem + (stage_elems > canon_elems ? stage_elems : canon_elems);  // [2][128] + dummy [2][128]

  typename Engine<T>::Acc acc;
  Engine<T>::zero(acc);
  kloop<T, true>(L + mat + (int64_t)row0 * ld, valid, nullptr, 0, ld, row0, sA, nullptr, acc, tid);

  T a[64];
  Engine<T>::canonical(acc, a, smem, wave, lane);

  const int r = 32 * wave + (lane & 31), g = lane >> 5;
  const bool rvalid = r < valid;
  const T* Hrow = H + mat + (int64_t)(row0 + r) * ld + row0;
  const T lam = damping ? damping[b] : T(0);
  // S = H_jj (+ damping on the diagonal) - acc ; identity padding outside the matrix
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int q0 = 8 * m + 4 * g;
    const int idx = 16 * (m >> 2) + 4 * (m & 3);
    T h[4] = {T(0), T(0), T(0), T(0)};
    if (rvalid && q0 < valid) {  // valid is a multiple of 2 (n = 6P), ld % 32 == 0: the 16-B chunk is in bounds
      if constexpr (sizeof(T) == 4) {
        const V v = *reinterpret_cast<const V*>(Hrow + q0);
        h[0] = v.x; h[1] = v.y; h[2] = v.z; h[3] = v.w;
      } else {
        const V v0 = *reinterpret_cast<const V*>(Hrow + q0);
        const V v1 = *reinterpret_cast<const V*>(Hrow + q0 + 2);
        h[0] = v0.x; h[1] = v0.y; h[2] = v1.x; h[3] = v1.y;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int q = q0 + t;
      T hv = h[t];
      if (q == r) {
        if (rvalid) {
          if (damping) hv = ellipsoidal ? hv + (lam * hv + damping_eps) : hv + lam;
        } else {
          hv = T(1);
        }
      } else if (!rvalid || q >= valid) {
        hv = T(0);
      }
      a[idx + t] = hv - a[idx + t];
    }
  }

  // ---- in-register right-looking Cholesky, one barrier per column ----
  int bad = 0;
  static_for<TILE>([&](auto ic) __attribute__((always_inline)) {
    constexpr int c = decltype(ic)::value;
    constexpr int G = (c >> 2) & 1;
    constexpr int I = 16 * (c >> 5) + 4 * ((c & 31) >> 3) + (c & 3);
    T* cb = colbuf + (c & 1) * TILE;
    // branch-free publish of column c: the non-owning lane group writes to a scratch half (keeps the
    // whole factorisation one basic block -- with branches LLVM sinks the FMAs and spills the LDS reads)
    cb[(g == G ? 0 : 2 * TILE) + r] = a[I];
    __syncthreads();
    T d = cb[c];
    if (!(d > T(0))) {
      if (bad == 0) bad = c + 1;
      d = T(1);
    }
    const T sq = t_sqrt(d);
    const T isq = t_rcp(sq);
    const T lrc = cb[r] * isq;  // L[r][c]
    const T s = lrc * isq;      // S[r][c] / d
#pragma unroll
    for (int m = (c >> 3); m < 16; ++m) {
      const V* vp = reinterpret_cast<const V*>(cb + 8 * m + 4 * g);
      T v[4];
      if constexpr (sizeof(T) == 4) {
        const V x = vp[0];
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
      } else {
        const V x0 = vp[0], x1 = vp[1];
        v[0] = x0.x; v[1] = x0.y; v[2] = x1.x; v[3] = x1.y;
      }
      const int idx = 16 * (m >> 2) + 4 * (m & 3);
      if (8 * m > c) {
#pragma unroll
        for (int t = 0; t < 4; ++t) a[idx + t] -= s * v[t];
      } else {  // chunk containing column c: only columns q > c
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const T se = (4 * g + t > (c & 7)) ? s : T(0);
          a[idx + t] -= se * v[t];
        }
      }
    }
    a[I] = (g == G) ? lrc : a[I];
    __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from hoisting later steps' LDS reads (spills)
  });
  if (bad != 0 && tid == 0 && info[b] == 0) info[b] = row0 + bad;  // every thread sees the same `bad`

  // ---- store L_jj (row major, zeros above the diagonal) and its transpose (for the solves) ----
  T* Lrow = L + mat + (int64_t)(row0 + r) * ld + row0;
  T* dT = diagT + ((int64_t)b * ntiles + j) * TILE * TILE;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int q0 = 8 * m + 4 * g;
    const int idx = 16 * (m >> 2) + 4 * (m & 3);
    T v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      v[t] = (q0 + t <= r) ? a[idx + t] : T(0);
      dT[(int64_t)(q0 + t) * TILE + r] = v[t];
    }
    if (rvalid && q0 < valid) {
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<V*>(Lrow + q0) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        *reinterpret_cast<V*>(Lrow + q0) = make_double2(v[0], v[1]);
        *reinterpret_cast<V*>(Lrow + q0 + 2) = make_double2(v[2], v[3]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// chol_offdiag: GEMM K-loop + in-register triangular solve  L_ij = (H_ij - sum) L_jj^-T
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256, sizeof(T) == 4 ? 2 : 1)
chol_offdiag_kernel(const T* __restrict__ H, T* __restrict__ L, const T* __restrict__ diagT, int n, int64_t ld,
                    int j, int ntiles, int nrow_tiles, int B) {
  using C = CT<T>;
  using V = typename C::V;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  // XCD-aware mapping: block id -> (problem, tile) so that all row tiles of one problem (which share
  // the L_j panel) run on the same XCD (blocks are dealt round-robin over the 8 XCDs).
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int b = (slot / nrow_tiles) * 8 + xcd;
  const int i = j + 1 + (slot % nrow_tiles);
  if (b >= B) return;  // batch padded to a multiple of 8 by the launcher (whole block exits)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t mat = (int64_t)b * ld * ld;
  const int col0 = j * TILE, row0 = i * TILE;
  const int validA = min(TILE, n - col0);  // == TILE (j is not the last tile)
  const int validB = min(TILE, n - row0);
  T* sA = smem;
  T* sB = smem + 128 * C::LDT;

  typename Engine<T>::Acc acc;
  Engine<T>::zero(acc);
  kloop<T, false>(L + mat + (int64_t)col0 * ld, validA, L + mat + (int64_t)row0 * ld, validB, ld, col0, sA, sB, acc,
                  tid);
  T p[64];
  Engine<T>::canonical(acc, p, smem, wave, lane);

  // L_jj^T tile -> LDS (LT[c][q] = L[col0+q][col0+c]); reciprocal diagonal
  __syncthreads();
  T* LT = smem;               // [128][128]
  T* dinv = smem + 128 * 128;  // [128]
  {
    const uint4* src = reinterpret_cast<const uint4*>(diagT + ((int64_t)b * ntiles + j) * TILE * TILE);
    uint4* dst = reinterpret_cast<uint4*>(LT);
    constexpr int NV = 128 * 128 * sizeof(T) / 16;
    for (int v = tid; v < NV; v += 256) dst[v] = src[v];
  }
  __syncthreads();
  if (tid < 128) dinv[tid] = t_rcp(LT[tid * 128 + tid]);

  const int r = 32 * wave + (lane & 31), g = lane >> 5;
  const bool rvalid = r < validB;
  const T* Hrow = H + mat + (int64_t)(row0 + r) * ld + col0;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int q0 = 8 * m + 4 * g;
    const int idx = 16 * (m >> 2) + 4 * (m & 3);
    T h[4] = {T(0), T(0), T(0), T(0)};
    if (rvalid) {
      if constexpr (sizeof(T) == 4) {
        const V v = *reinterpret_cast<const V*>(Hrow + q0);
        h[0] = v.x; h[1] = v.y; h[2] = v.z; h[3] = v.w;
      } else {
        const V v0 = *reinterpret_cast<const V*>(Hrow + q0);
        const V v1 = *reinterpret_cast<const V*>(Hrow + q0 + 2);
        h[0] = v0.x; h[1] = v0.y; h[2] = v1.x; h[3] = v1.y;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) p[idx + t] = h[t] - p[idx + t];
  }
  __syncthreads();  // dinv visible

  // ---- X L_jj^T = P, right-looking over columns; each row lives in the lane pair (l, l^32) ----
  const T own0 = g == 0 ? T(1) : T(0), own1 = T(1) - own0;
  static_for<TILE>([&](auto ic) __attribute__((always_inline)) {
    constexpr int c = decltype(ic)::value;
    constexpr int G = (c >> 2) & 1;
    constexpr int I = 16 * (c >> 5) + 4 * ((c & 31) >> 3) + (c & 3);
    // arithmetic mask instead of a select: LLVM turns `cond ? p*dinv[c] : 0` back into a branch
    // around the LDS load, and the branches let it sink the FMA chains and spill every LT read
    const T mine = p[I] * dinv[c] * (G == 0 ? own0 : own1);
    const T x = pair_sum(mine);
    const T* ltc = LT + c * 128;
#pragma unroll
    for (int m = (c >> 3); m < 16; ++m) {
      const V* vp = reinterpret_cast<const V*>(ltc + 8 * m + 4 * g);
      T v[4];
      if constexpr (sizeof(T) == 4) {
        const V y = vp[0];
        v[0] = y.x; v[1] = y.y; v[2] = y.z; v[3] = y.w;
      } else {
        const V y0 = vp[0], y1 = vp[1];
        v[0] = y0.x; v[1] = y0.y; v[2] = y1.x; v[3] = y1.y;
      }
      const int idx = 16 * (m >> 2) + 4 * (m & 3);
#pragma unroll
      for (int t = 0; t < 4; ++t) p[idx + t] -= x * v[t];  // LT[c][q] == 0 for q < c
    }
    p[I] = (g == G) ? x : p[I];
    __builtin_amdgcn_sched_barrier(0);
  });

  if (rvalid) {
    T* Lrow = L + mat + (int64_t)(row0 + r) * ld + col0;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int q0 = 8 * m + 4 * g;
      const int idx = 16 * (m >> 2) + 4 * (m & 3);
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<V*>(Lrow + q0) = make_float4(p[idx], p[idx + 1], p[idx + 2], p[idx + 3]);
      } else {
        *reinterpret_cast<V*>(Lrow + q0) = make_double2(p[idx], p[idx + 1]);
        *reinterpret_cast<V*>(Lrow + q0 + 2) = make_double2(p[idx + 2], p[idx + 3]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// triangular solves with one right-hand side per problem (v1: one workgroup per problem)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T>
__global__ void __launch_bounds__(256)
chol_solve_kernel(const T* __restrict__ L, const T* __restrict__ diagT, const T* __restrict__ rhs, T* __restrict__ x,
                  int n, int64_t ld, int64_t ldv, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int npad = ntiles * TILE;
  T* tile = smem;                   // [128][128]
  T* y = smem + 128 * 128;          // [npad]
  T* tbuf = y + npad;               // [128]
  T* cur = tbuf + 128;              // [2]
  const T* Lb = L + (int64_t)b * ld * ld;
  const T* dTb = diagT + (int64_t)b * ntiles * TILE * TILE;
  for (int k = tid; k < npad; k += 256) y[k] = k < n ? rhs[(int64_t)b * ldv + k] : T(0);
  __syncthreads();

  // ---- forward: L y = g ----
  for (int jb = 0; jb < ntiles; ++jb) {
    const int row0 = jb * TILE, K = row0;
    const int valid = min(TILE, n - row0);
    // t[r] = sum_k L[row0+r][k] y[k], one wave per row (coalesced along k)
    for (int rr = wave; rr < TILE; rr += 4) {
      T s = T(0);
      if (rr < valid) {
        const T* Lr = Lb + (int64_t)(row0 + rr) * ld;
        for (int k = lane; k < K; k += 64) s += Lr[k] * y[k];
      }
      s = wave_sum(s);
      if (lane == 0) tbuf[rr] = s;
    }
    {  // LT tile of this diagonal block
      const uint4* src = reinterpret_cast<const uint4*>(dTb + (int64_t)jb * TILE * TILE);
      uint4* dst = reinterpret_cast<uint4*>(tile);
      constexpr int NV = 128 * 128 * sizeof(T) / 16;
      for (int v = tid; v < NV; v += 256) dst[v] = src[v];
    }
    __syncthreads();
    T rq = T(0);
    if (tid < TILE) rq = y[row0 + tid] - tbuf[tid];
    for (int c = 0; c < TILE; ++c) {
      if (tid == c) cur[c & 1] = rq / tile[c * 128 + c];
      __syncthreads();
      const T yc = cur[c & 1];
      if (tid == c) rq = yc;
      else if (tid > c && tid < TILE) rq -= yc * tile[c * 128 + tid];  // LT[c][q] = L[q][c]
    }
    if (tid < TILE) y[row0 + tid] = rq;
    __syncthreads();
  }

  // ---- backward: L^T x = y ----
  for (int jb = ntiles - 1; jb >= 0; --jb) {
    const int row0 = jb * TILE;
    const int valid = min(TILE, n - row0);
    {
      const uint4* src = reinterpret_cast<const uint4*>(dTb + (int64_t)jb * TILE * TILE);
      uint4* dst = reinterpret_cast<uint4*>(tile);
      constexpr int NV = 128 * 128 * sizeof(T) / 16;
      for (int v = tid; v < NV; v += 256) dst[v] = src[v];
    }
    __syncthreads();
    // x_c = (z_c - sum_{q>c} L[q][c] x_q) / L[c][c] ; column c of L == row c of LT: thread q holds x_q
    // right-looking in reverse: after x_c is final, z_q -= L[c][q] x_c for q < c, L[c][q] = LT[q][c]
    T zq = T(0);
    if (tid < TILE) zq = y[row0 + tid];
    for (int c = TILE - 1; c >= 0; --c) {
      if (tid == c) cur[c & 1] = zq / tile[c * 128 + c];
      __syncthreads();
      const T xc = cur[c & 1];
      if (tid == c) zq = xc;
      else if (tid < c) zq -= xc * tile[tid * 128 + c];
    }
    if (tid < TILE) y[row0 + tid] = zq;  // now x for this block
    __syncthreads();
    // z[0:row0] -= L[row0: row0+valid, 0:row0]^T x_block   (thread per column, coalesced rows)
    for (int k = tid; k < row0; k += 256) {
      T s = T(0);
      for (int rr = 0; rr < valid; ++rr) s += Lb[(int64_t)(row0 + rr) * ld + k] * y[row0 + rr];
      y[k] -= s;
    }
    __syncthreads();
  }
  for (int k = tid; k < n; k += 256) x[(int64_t)b * ldv + k] = y[k];
}

// ------------------------------------------------------------------------------------------------
// small helpers: diagonal extraction, LM accept test
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void diag_kernel(const T* __restrict__ H, int64_t ld, int n, T* __restrict__ d, int64_t ldv) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[(int64_t)b * ldv + i] = H[(int64_t)b * ld * ld + (int64_t)i * ld + i];
}

template <typename T>
__global__ void __launch_bounds__(64)
lm_accept_kernel(const T* __restrict__ delta, const T* __restrict__ g, int64_t ldv, const T* __restrict__ H,
                 int64_t ld, int n, T* __restrict__ damping, const T* __restrict__ prev_err,
                 const T* __restrict__ new_err, int ellipsoidal, T accept, T down, T up, uint8_t* __restrict__ reject) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const T lam = damping[b];
  T s = T(0);
  for (int i = lane; i < n; i += 64) {
    const T dl = delta[(int64_t)b * ldv + i];
    const T D = ellipsoidal ? H[(int64_t)b * ld * ld + (int64_t)i * ld + i] * lam : lam;
    s += dl * (D * dl + g[(int64_t)b * ldv + i]);
  }
  s = wave_sum(s);
  if (lane == 0) {
    const T den = s / T(2);
    const T rho = (prev_err[b] - new_err[b]) / den;
    const bool rej = rho <= accept;
    T nl = rej ? lam * up : lam / down;
    nl = nl < T(1.0e-7) ? T(1.0e-7) : (nl > T(1.0e7) ? T(1.0e7) : nl);
    damping[b] = nl;
    reject[b] = rej ? 1 : 0;
  }
}

template <typename T>
static size_t diag_smem() {
  size_t stage = (size_t)128 * CT<T>::LDT * sizeof(T);
  size_t canon = Engine<T>::canonical_lds_bytes;
  return (stage > canon ? stage : canon) + 4 * 128 * sizeof(T);
}
template <typename T>
static size_t offdiag_smem() {
  size_t stage = (size_t)2 * 128 * CT<T>::LDT * sizeof(T);
  size_t canon = Engine<T>::canonical_lds_bytes;
  size_t lt = (size_t)(128 * 128 + 128) * sizeof(T);
  size_t m = stage > canon ? stage : canon;
  return m > lt ? m : lt;
}

template <typename T>
static int factor_impl(const void* H, int64_t ld, int n, int B, const void* damping, int ellipsoidal, double eps,
                       void* L, void* diagT, int32_t* info, hipStream_t st) {
  const int ntiles = (n + TILE - 1) / TILE;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_diag_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)diag_smem<T>());
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_kernel<T>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)offdiag_smem<T>());
    attr_set = true;
  }
  hipMemsetAsync(info, 0, sizeof(int32_t) * (size_t)B, st);
  const int Bpad = (B + 7) / 8 * 8;
  for (int j = 0; j < ntiles; ++j) {
    hipLaunchKernelGGL(chol_diag_kernel<T>, dim3(B), dim3(256), diag_smem<T>(), st, (const T*)H, (T*)L, (T*)diagT,
                       (const T*)damping, ellipsoidal, (T)eps, info, n, ld, j, ntiles);
    const int nrt = ntiles - 1 - j;
    if (nrt > 0)
      hipLaunchKernelGGL(chol_offdiag_kernel<T>, dim3(Bpad * nrt), dim3(256), offdiag_smem<T>(), st, (const T*)H,
                         (T*)L, (const T*)diagT, n, ld, j, ntiles, nrt, B);
  }
  return check_launch("thx_chol_factor");
}

}  // namespace thx

using namespace thx;

extern "C" {

int thx_chol_factor(const void* H, int64_t ld, int32_t n, int32_t B, const void* damping, int ellipsoidal,
                    double damping_eps, void* L, void* diagT, int32_t* info, int dtype, void* stream) {
  if (!H || !L || !diagT || !info) return fail("thx_chol_factor: null pointer");
  if (n <= 0 || B <= 0 || ld < n || (ld % 32) != 0) return fail("thx_chol_factor: need n>0, B>0, ld>=n, ld%32==0");
  if (n % 2) return fail("thx_chol_factor: n must be even");
  THX_DISPATCH(dtype,
               return factor_impl<float>(H, ld, n, B, damping, ellipsoidal, damping_eps, L, diagT, info,
                                         as_stream(stream)),
               return factor_impl<double>(H, ld, n, B, damping, ellipsoidal, damping_eps, L, diagT, info,
                                          as_stream(stream)));
  return 0;
}

int thx_chol_solve(const void* L, int64_t ld, int32_t n, int32_t B, const void* diagT, const void* rhs, void* x,
                   int64_t ldv, int dtype, void* stream) {
  if (!L || !diagT || !rhs || !x) return fail("thx_chol_solve: null pointer");
  if (n <= 0 || B <= 0 || ld < n || ldv < n) return fail("thx_chol_solve: bad sizes");
  const int ntiles = (n + TILE - 1) / TILE;
  THX_DISPATCH(
      dtype,
      {
        const size_t sm = (size_t)(128 * 128 + ntiles * TILE + 128 + 2) * sizeof(float);
        hipFuncSetAttribute(reinterpret_cast<const void*>(chol_solve_kernel<float>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipLaunchKernelGGL(chol_solve_kernel<float>, dim3(B), dim3(256), sm, as_stream(stream), (const float*)L,
                           (const float*)diagT, (const float*)rhs, (float*)x, n, ld, ldv, ntiles);
      },
      {
        const size_t sm = (size_t)(128 * 128 + ntiles * TILE + 128 + 2) * sizeof(double);
        if (sm > 160 * 1024) return fail("thx_chol_solve: n too large for the f64 LDS plan");
        hipFuncSetAttribute(reinterpret_cast<const void*>(chol_solve_kernel<double>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipLaunchKernelGGL(chol_solve_kernel<double>, dim3(B), dim3(256), sm, as_stream(stream), (const double*)L,
                           (const double*)diagT, (const double*)rhs, (double*)x, n, ld, ldv, ntiles);
      });
  return check_launch("thx_chol_solve");
}

int thx_diag(const void* H, int64_t ld, int32_t n, int32_t B, void* d, int64_t ldv, int dtype, void* stream) {
  if (!H || !d || n <= 0 || B <= 0) return fail("thx_diag: bad args");
  dim3 grid((n + 255) / 256, B), block(256);
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(diag_kernel<float>, grid, block, 0, as_stream(stream), (const float*)H, ld, n,
                                  (float*)d, ldv),
               hipLaunchKernelGGL(diag_kernel<double>, grid, block, 0, as_stream(stream), (const double*)H, ld, n,
                                  (double*)d, ldv));
  return check_launch("thx_diag");
}

int thx_lm_accept(const void* delta, const void* g, int64_t ldv, const void* H, int64_t ld, int32_t n, int32_t B,
                  void* damping, const void* prev_err, const void* new_err, int ellipsoidal, double accept,
                  double down_ratio, double up_ratio, uint8_t* reject, int dtype, void* stream) {
  if (!delta || !g || !damping || !prev_err || !new_err || !reject || (ellipsoidal && !H))
    return fail("thx_lm_accept: null pointer");
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(lm_accept_kernel<float>, dim3(B), dim3(64), 0, as_stream(stream), (const float*)delta,
                                  (const float*)g, ldv, (const float*)H, ld, n, (float*)damping,
                                  (const float*)prev_err, (const float*)new_err, ellipsoidal, (float)accept,
                                  (float)down_ratio, (float)up_ratio, reject),
               hipLaunchKernelGGL(lm_accept_kernel<double>, dim3(B), dim3(64), 0, as_stream(stream),
                                  (const double*)delta, (const double*)g, ldv, (const double*)H, ld, n,
                                  (double*)damping, (const double*)prev_err, (const double*)new_err, ellipsoidal,
                                  accept, down_ratio, up_ratio, reject));
  return check_launch("thx_lm_accept");
}

}  // extern "C"
