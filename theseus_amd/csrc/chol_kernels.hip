// Batched damped dense Cholesky (factor + solves) for gfx950, one problem per workgroup-chain.
//
// Replaces DenseSolver._apply_damping + CholeskyDenseSolver._solve_sytem
// (theseus/optimizer/linear/dense_solver.py:38-64,159-161) for a batch of B SPD matrices of order n.
//
// Algorithm: tiled LEFT-LOOKING Cholesky with 128x128 tiles, one kernel launch pair per block
// column j (the batch supplies the parallelism: B x (N-j) workgroups per launch, no inter-workgroup
// communication inside a launch):
//   chol_diag(j)    : S = H_jj + damping - L_j,0:j L_j,0:j^T   (MFMA K-loop; the same pass over the
//                     panel also forms L_j,0:j y_0:j for the fused forward substitution)
//                     L_jj = chol(S)                            (register-resident, row per lane pair)
//                     panel M_j: 32x32 diagonal sub-blocks W_ss = L_ss^-1 (inverted in fp64), strictly
//                     lower sub-blocks -L_st;  y_j = L_jj^-1 (g_j - L_j,0:j y)  (blocked, via M_j)
//   chol_offdiag(j) : P = H_ij - L_i,0:j L_j,0:j^T              (MFMA K-loop)
//                     L_ij = P L_jj^-T as a BLOCKED substitution on the matrix cores:
//                       for s = 0..3:  P_s += sum_{t<s} X_t (-L_st)^T ;  X_s = P_s W_ss^T
//                     The accumulator registers are fed straight back as the MFMA B operand (the
//                     k index of an MFMA is just a pairing of columns, and the pairing the C/D
//                     layout gives is as good as any), A comes from the panel in LDS: no data
//                     movement, 160 MFMAs per wave instead of 128 dependent VALU/LDS steps.
// Left-looking means every tile of L is written exactly once and the trailing matrix is never
// re-read: per problem the K-loops stream  sum_j (N-j) * 2*128*(128 j)  elements.
// Only 32x32 triangles are ever inverted (in fp64, rounded once): the substitution across sub-blocks
// uses L itself, so the result has the error profile of a blocked TRSM, not of a full inverse.
//
// MFMA mapping (f32: v_mfma_f32_32x32x2_f32, f64: v_mfma_f64_16x16x4_f64): a 256-thread workgroup is
// 4 waves; wave w owns tile rows [32w, 32w+32) x all 128 columns and computes the TRANSPOSED product
// block D = L_j-rows * L_i-rows^T, so that a lane holds ONE matrix row (f32: row 32w + (lane&31),
// columns with ((c>>2)&1) == lane>>5; f64: row 32w + 16h + (lane&15), columns c = lane>>4 mod 4).
//
// Solves: L y = g is fused into the factorisation (above); L^T x = y is one HBM-bound pass over L
// (chol_bwd_kernel).  A stand-alone forward kernel serves solves with a cached factor (implicit
// backward pass).
#include "common.cuh"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>
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
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <typename T>
struct CT;
template <>
struct CT<float> {
  static constexpr int KB = 32, LDT = 36, VEC = 4, LDM = 132, LDB = 36;
  using V = float4;
};
template <>
struct CT<double> {
  // k-chunks of 16 columns, two of them in flight (kloop_f: AHEAD).  32-column chunks -- KB = 32, LDT = 34: everything below is
  // written for either -- halve the barriers and staging round trips per flop and measured SLOWER: factor 111.2 vs 108.8 ms at
  // n = 1536 / batch 4096, 27.8 vs 26.3 ms at batch 1024, 44.7 vs 43.6 ms at n = 3072 / batch 256 on one box
  // (profiles/r3/j_ab_f64_kchunk32_vs_16.txt): the fp64 K-loop is not barrier-bound; with 70 KB of staging per workgroup the two
  // workgroups of a CU leave no LDS slack and one 32-column chunk in flight hides less latency than two 16-column ones.
  static constexpr int KB = 16, LDT = 18, VEC = 2, LDM = 130, LDB = 34;
  using V = double2;
};

// The diagonal tile in LDS (chol_diag): only its ten lower 32x32 sub-blocks, each stored row-major with row stride LDB
// (= 4 banks mod 32, like the 128-wide rows they replace: 46 KB instead of 68 KB in fp32 -> three workgroups per CU).
template <typename T>
__device__ __forceinline__ constexpr int tblk(int u, int v) {
  return (u * (u + 1) / 2 + v) * 32 * CT<T>::LDB;
}

// 1/sqrt(d) from the hardware estimate + Newton steps (~1 ulp): the pivot scaling of the in-register
// Cholesky sits on a 128-step latency chain, a correctly rounded sqrt + division is ~40 dependent
// instructions, this is ~8.  L[c][c] = d * isq and L[r][c] = S[r][c] * isq stay mutually consistent.
__device__ __forceinline__ float t_rsqrt(float d) {
  float y = __builtin_amdgcn_rsqf(d);
  return y * (1.5f - 0.5f * d * y * y);
}
__device__ __forceinline__ double t_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  return y * (1.5 - 0.5 * d * y * y);
}

// lane broadcast through SGPRs (v_readlane_b32; `l` is wave uniform)
__device__ __forceinline__ float bcast(float x, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}
__device__ __forceinline__ double bcast(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                          __builtin_amdgcn_readlane(__double2loint(x), l));
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
#ifdef THX_KLOOP_PIPE
  // EXPERIMENT (round 4): the fragment reads software-pipelined by hand -- hipcc emits "ds_read_b128; s_waitcnt lgkmcnt(0); 4 MFMAs"
  // per A fragment; here the read of fragment n + 1 is issued before the MFMAs of fragment n (a two-deep register ring, the four
  // B fragments of the chunk loaded up front) and the order is pinned with sched_group_barrier (0x100 = DS read, 0x008 = MFMA).
  static __device__ __forceinline__ void chunk(const float* sA, const float* sBw, Acc& acc, int lane) {
    const int rl = lane & 31, g = lane >> 5;
    float4 fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[ks] = *reinterpret_cast<const float4*>(sBw + rl * 36 + 8 * ks + 4 * g);
    auto ld = [&](int n) __attribute__((always_inline)) {   // fragment n = (ks, cb) = (n / 4, n % 4)
      return *reinterpret_cast<const float4*>(sA + (32 * (n & 3) + rl) * 36 + 8 * (n >> 2) + 4 * g);
    };
    float4 ring[2];
    ring[0] = ld(0);
    __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      if (n + 1 < 16) ring[(n + 1) & 1] = ld(n + 1);
      const float4 fa = ring[n & 1];
      const float4 b = fb[n >> 2];
      const int cb = n & 3;
      acc.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, b.x, acc.v[cb], 0, 0, 0);
      acc.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, b.y, acc.v[cb], 0, 0, 0);
      acc.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, b.z, acc.v[cb], 0, 0, 0);
      acc.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, b.w, acc.v[cb], 0, 0, 0);
      if (n + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  }
#else
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
#endif

  // SYRK of the diagonal tile on the 36 lower 16x16 blocks of its 8x8 block grid, nine per wave: wave g owns block
  // rows 4+g (5+g blocks) and 3-g (4-g blocks) -- equal MFMA counts on all four SIMDs, 56 % of the full tile.
  // v_mfma_f32_16x16x4_f32.  Staged rows have stride SYRK_LDT = 40 words and lane (r, kq) reads the two 16-byte
  // pieces at words 4kq and 16+4kq (an MFMA's k index is only a pairing of columns): the one (stride, offsets)
  // combination for which the ds_read_b128 of a 16x16 fragment is bank-conflict free (stride 36: 32 % conflicts).
  static constexpr int SYRK_LDT = 40;
  static constexpr int SYRK_STAGE = 128 * SYRK_LDT;   // elements of the SYRK K-loop's staging buffer
  using Sy = f32x4;
  template <int G>
  static __device__ __forceinline__ void syrk36(const float* sA, f32x4* acc, int lane) {
    constexpr int UH = 4 + G, UL = 3 - G;
    const int o = (lane & 15) * 40 + 4 * (lane >> 4);
    float fbh[8], fbl[8];
    {
      const float4 a = *reinterpret_cast<const float4*>(sA + 16 * UH * 40 + o), b = *reinterpret_cast<const float4*>(sA + 16 * UH * 40 + o + 16);
      const float4 c = *reinterpret_cast<const float4*>(sA + 16 * UL * 40 + o), d = *reinterpret_cast<const float4*>(sA + 16 * UL * 40 + o + 16);
      fbh[0] = a.x; fbh[1] = a.y; fbh[2] = a.z; fbh[3] = a.w; fbh[4] = b.x; fbh[5] = b.y; fbh[6] = b.z; fbh[7] = b.w;
      fbl[0] = c.x; fbl[1] = c.y; fbl[2] = c.z; fbl[3] = c.w; fbl[4] = d.x; fbl[5] = d.y; fbl[6] = d.z; fbl[7] = d.w;
    }
#pragma unroll
    for (int v = 0; v <= UH; ++v) {
      float fa[8];
      if (v == UH) {
#pragma unroll
        for (int m = 0; m < 8; ++m) fa[m] = fbh[m];
      } else if (v == UL) {
#pragma unroll
        for (int m = 0; m < 8; ++m) fa[m] = fbl[m];
      } else {
        const float4 a = *reinterpret_cast<const float4*>(sA + 16 * v * 40 + o), b = *reinterpret_cast<const float4*>(sA + 16 * v * 40 + o + 16);
        fa[0] = a.x; fa[1] = a.y; fa[2] = a.z; fa[3] = a.w; fa[4] = b.x; fa[5] = b.y; fa[6] = b.z; fa[7] = b.w;
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m], fbh[m], acc[v], 0, 0, 0);
      if (v <= UL) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[UH + 1 + v] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m], fbl[m], acc[UH + 1 + v], 0, 0, 0);
      }
    }
  }
  // The same nine blocks of H_jj, global -> registers BEFORE the K-loop (row index clamped into the matrix: the caller
  // masks by value), so that the H tile costs no exposed round trips after it.
  template <int G>
  static __device__ __forceinline__ void syrk36_prefetch(const float* __restrict__ Hjj, int64_t ld, int valid, float4* hp,
                                                        int lane) {
    constexpr int UH = 4 + G, UL = 3 - G;
    auto ldb = [&](int u, int v) __attribute__((always_inline)) -> float4 {
      const int r = min(16 * u + (lane & 15), valid - 1);
      return *reinterpret_cast<const float4*>(Hjj + (int64_t)r * ld + 16 * v + 4 * (lane >> 4));
    };
#pragma unroll
    for (int v = 0; v <= UH; ++v) hp[v] = ldb(UH, v);
#pragma unroll
    for (int v = 0; v <= UL; ++v) hp[UH + 1 + v] = ldb(UL, v);
  }
  // tile(u, v) <- S = H (+ damping on the diagonal) - acc; rows / columns outside the matrix: identity
  template <int G>
  static __device__ __forceinline__ void syrk36_store(float* tile, const float4* hp, const f32x4* acc, int lane, int valid,
                                                      bool damp, float lam, int ellipsoidal, float eps) {
    constexpr int UH = 4 + G, UL = 3 - G;
    auto st = [&](int u, int v, const float4& h, const f32x4& a) __attribute__((always_inline)) {
      const int r = 16 * u + (lane & 15), c0 = 16 * v + 4 * (lane >> 4);
      float hv[4] = {h.x, h.y, h.z, h.w}, o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c0 + k;
        float x = hv[k];
        if (u == v && r == c && damp) x = ellipsoidal ? x + (lam * x + eps) : x + lam;
        x -= a[k];
        if (r >= valid || c >= valid) x = (r == c) ? 1.f : 0.f;
        o[k] = x;
      }
      *reinterpret_cast<float4*>(tile + tblk<float>(u >> 1, v >> 1) + (r & 31) * 36 + (c0 & 31)) =
          make_float4(o[0], o[1], o[2], o[3]);
    };
#pragma unroll
    for (int v = 0; v <= UH; ++v) st(UH, v, hp[v], acc[v]);
#pragma unroll
    for (int v = 0; v <= UL; ++v) st(UL, v, hp[UH + 1 + v], acc[UH + 1 + v]);
  }
  // ---- 32x32 block helpers for the diagonal-tile factorisation (operands: sub-blocks of the LDS tile, row stride 36).
  //      Blk D[m][n]: a lane holds ONE row n = lane&31 of the block, register rho <-> column m = 8(rho>>2) + 4g + (rho&3)
  using Blk = f32x16;
  static __device__ __forceinline__ void blk_zero(Blk& d) {
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
  }
  static __device__ __forceinline__ void blk_sub(Blk& d, const Blk& a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] -= a[i];
  }
  static __device__ __forceinline__ void blk_load(Blk& d, const float* blk, int lane) {
    const float* p = blk + (lane & 31) * 36 + 4 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(p + 8 * q);
      d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
    }
  }
  static __device__ __forceinline__ void blk_store(const Blk& d, float* blk, int lane, float sign) {
    float* p = blk + (lane & 31) * 36 + 4 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(p + 8 * q) =
          make_float4(sign * d[4 * q], sign * d[4 * q + 1], sign * d[4 * q + 2], sign * d[4 * q + 3]);
  }
  // D[m][n] += sum_k (asign * A[m][k]) * B[n][k]   (A, B: 32x32 blocks in LDS, rows m / n)
  static __device__ __forceinline__ void blk_mma(const float* Ablk, const float* Bblk, Blk& d, int lane, float asign) {
    const int o = (lane & 31) * 36 + 4 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 fa = *reinterpret_cast<const float4*>(Ablk + o + 8 * q);
      const float4 fb = *reinterpret_cast<const float4*>(Bblk + o + 8 * q);
      d = __builtin_amdgcn_mfma_f32_32x32x2f32(asign * fa.x, fb.x, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x2f32(asign * fa.y, fb.y, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x2f32(asign * fa.z, fb.z, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x2f32(asign * fa.w, fb.w, d, 0, 0, 0);
    }
  }

  // ---- register-resident blocks (chol_potrf_kernel: the whole diagonal tile lives in ONE wave's registers) ----
  // D[m][n] += sum_k (asign * A[m][k]) * B[n][k] with A, B AND D in the C/D layout (lane = the block's own row, registers =
  // columns): an MFMA's k index is only a pairing of columns, and two blocks in this layout pair theirs identically
  // (register rho of lane group g <-> column 8(rho>>2) + 4g + (rho&3)).  No LDS, no data movement.
  static __device__ __forceinline__ void blk_mma_rr(const Blk& A, const Blk& B, Blk& d, float asign) {
#pragma unroll
    for (int i = 0; i < 16; ++i) d = __builtin_amdgcn_mfma_f32_32x32x2f32(asign * A[i], B[i], d, 0, 0, 0);
  }
  static __device__ __forceinline__ void blk_neg(Blk& d) {
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = -d[i];
  }
  // 32x32 block of a row-major global matrix -> registers; rows >= rv / columns >= cv (outside the matrix): identity on a
  // diagonal block, zero elsewhere
  static __device__ __forceinline__ void blk_load_global(Blk& d, const float* blk, const float* safe, int64_t ld, int rv, int cv,
                                                         bool diag, int lane) {
    const int r = lane & 31, g = lane >> 5;
    // (a sub-block wholly outside the matrix may lie outside the FRAME too: its lanes read ``safe`` -- 32 valid elements)
    const float* p = (r < rv ? blk + (int64_t)r * ld : safe) + 4 * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(p + 8 * q);
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = 8 * q + 4 * g + k;
        d[4 * q + k] = (r < rv && c < cv) ? e[k] : ((diag && r == c) ? 1.f : 0.f);
      }
    }
  }
  // registers -> global (rows < rv only), scaled by sign
  static __device__ __forceinline__ void blk_store_global(const Blk& d, float* blk, int64_t ld, int rv, int lane, float sign) {
    const int r = lane & 31, g = lane >> 5;
    if (r < rv) {
      float* p = blk + (int64_t)r * ld + 4 * g;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(p + 8 * q) =
            make_float4(sign * d[4 * q], sign * d[4 * q + 1], sign * d[4 * q + 2], sign * d[4 * q + 3]);
    }
  }
  // rows of a 32-vector a lane is responsible for: NR of them, row blk_row(lane, i); the lanes with row_owner() write
  static constexpr int NR = 1;
  static __device__ __forceinline__ int blk_row(int lane, int) { return lane & 31; }
  static __device__ __forceinline__ bool row_owner(int lane) { return lane < 32; }
  // out[0] = sum_c X[row][c] * vec[c] for this lane's row (vec: 32 values in LDS); valid in every lane
  static __device__ __forceinline__ void blk_rowdot(const Blk& X, const float* vec, int lane, float (&out)[1]) {
    const int g = lane >> 5;
    // (explicit FMA chain: the function is inlined into two kernels whose results must agree bit for bit -- fwd_diag_block --
    //  and the compiler's contraction of a * b + c * d + ... depends on the surrounding code)
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(vec + 8 * q + 4 * g);
      s = __builtin_fmaf(X[4 * q], v.x, s);
      s = __builtin_fmaf(X[4 * q + 1], v.y, s);
      s = __builtin_fmaf(X[4 * q + 2], v.z, s);
      s = __builtin_fmaf(X[4 * q + 3], v.w, s);
    }
    out[0] = s + __shfl_xor(s, 32);
  }
  // S = H (+ damping on the diagonal) - acc of the 36 lower 16x16 blocks -> the diagonal tile's place in the GLOBAL factor
  // (rows inside the matrix only; chol_potrf_kernel pads on load)
  template <int G>
  static __device__ __forceinline__ void syrk36_store_global(float* Lt, int64_t ld, const float4* hp, const f32x4* acc, int lane,
                                                             int valid, bool damp, float lam, int ellipsoidal, float eps) {
    constexpr int UH = 4 + G, UL = 3 - G;
    auto st = [&](int u, int v, const float4& h, const f32x4& a) __attribute__((always_inline)) {
      const int r = 16 * u + (lane & 15), c0 = 16 * v + 4 * (lane >> 4);
      if (r >= valid) return;
      float hv[4] = {h.x, h.y, h.z, h.w}, o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float x = hv[k];
        if (u == v && r == c0 + k && damp) x = ellipsoidal ? x + (lam * x + eps) : x + lam;
        o[k] = x - a[k];
      }
      *reinterpret_cast<float4*>(Lt + (int64_t)r * ld + c0) = make_float4(o[0], o[1], o[2], o[3]);
    };
#pragma unroll
    for (int v = 0; v <= UH; ++v) st(UH, v, hp[v], acc[v]);
#pragma unroll
    for (int v = 0; v <= UL; ++v) st(UL, v, hp[UH + 1 + v], acc[UH + 1 + v]);
  }
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
    constexpr int LDT = CT<double>::LDT;
#pragma unroll
    for (int ks = 0; ks < CT<double>::KB / 8; ++ks) {
      double2 fb[2];
      fb[0] = *reinterpret_cast<const double2*>(sBw + rl * LDT + 8 * ks + 2 * kq);
      fb[1] = *reinterpret_cast<const double2*>(sBw + (16 + rl) * LDT + 8 * ks + 2 * kq);
#pragma unroll
      for (int cb = 0; cb < 8; ++cb) {
        const double2 fa = *reinterpret_cast<const double2*>(sA + (16 * cb + rl) * LDT + 8 * ks + 2 * kq);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          acc.v[h][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa.x, fb[h].x, acc.v[h][cb], 0, 0, 0);
          acc.v[h][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa.y, fb[h].y, acc.v[h][cb], 0, 0, 0);
        }
      }
    }
  }

  // ---- 32x32 block helpers (see Engine<float>): Blk.v[mh][nh][rho] <-> row n = 16nh + (lane&15),
  //      column m = 16mh + (lane>>4) + 4rho; LDS row stride 34
  struct Blk {
    f64x4 v[2][2];
  };
  // see Engine<float>::syrk36; staged rows have stride 20 doubles (40 words), a k-chunk is 16 wide, lane (r, kq) reads
  // doubles 2kq, 2kq+1 and 8+2kq, 9+2kq (words 4kq and 16+4kq): conflict free
  static constexpr int SYRK_LDT = 20;
  // (a 32-column k-chunk would be staged as TWO 16-wide sub-chunks of 128 x SYRK_LDT -- kloop_f's SPLIT16 layout -- so that each
  //  keeps the conflict-free stride; syrk36 runs once per sub-chunk)
  static constexpr int SYRK_SUBS = CT<double>::KB / 16;
  static constexpr int SYRK_STAGE = SYRK_SUBS * 128 * SYRK_LDT;
  using Sy = f64x4;
  template <int G>
  static __device__ __forceinline__ void syrk36(const double* sA, f64x4* acc, int lane) {
#pragma unroll
    for (int h = 0; h < SYRK_SUBS; ++h) syrk36_sub<G>(sA + h * 128 * SYRK_LDT, acc, lane);
  }
  template <int G>
  static __device__ __forceinline__ void syrk36_sub(const double* sA, f64x4* acc, int lane) {
    constexpr int UH = 4 + G, UL = 3 - G;
    const int o = (lane & 15) * 20 + 2 * (lane >> 4);
    double fbh[4], fbl[4];
    {
      const double2 a = *reinterpret_cast<const double2*>(sA + 16 * UH * 20 + o), b = *reinterpret_cast<const double2*>(sA + 16 * UH * 20 + o + 8);
      const double2 c = *reinterpret_cast<const double2*>(sA + 16 * UL * 20 + o), d = *reinterpret_cast<const double2*>(sA + 16 * UL * 20 + o + 8);
      fbh[0] = a.x; fbh[1] = a.y; fbh[2] = b.x; fbh[3] = b.y;
      fbl[0] = c.x; fbl[1] = c.y; fbl[2] = d.x; fbl[3] = d.y;
    }
#pragma unroll
    for (int v = 0; v <= UH; ++v) {
      double fa[4];
      if (v == UH) {
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[m] = fbh[m];
      } else if (v == UL) {
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[m] = fbl[m];
      } else {
        const double2 a = *reinterpret_cast<const double2*>(sA + 16 * v * 20 + o), b = *reinterpret_cast<const double2*>(sA + 16 * v * 20 + o + 8);
        fa[0] = a.x; fa[1] = a.y; fa[2] = b.x; fa[3] = b.y;
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[v] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[m], fbh[m], acc[v], 0, 0, 0);
      if (v <= UL) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[UH + 1 + v] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[m], fbl[m], acc[UH + 1 + v], 0, 0, 0);
      }
    }
  }
  // H_jj blocks global -> registers before the K-loop / S = H (+ damping) - acc -> LDS afterwards (see Engine<float>)
  template <int G>
  static __device__ __forceinline__ void syrk36_prefetch(const double* __restrict__ Hjj, int64_t ld, int valid, f64x4* hp,
                                                        int lane) {
    constexpr int UH = 4 + G, UL = 3 - G;
    auto ldb = [&](int u, int v) __attribute__((always_inline)) -> f64x4 {
      const double* p = Hjj + (int64_t)min(16 * u + (lane & 15), valid - 1) * ld + 16 * v + (lane >> 4);
      f64x4 h;
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) h[rho] = p[4 * rho];
      return h;
    };
#pragma unroll
    for (int v = 0; v <= UH; ++v) hp[v] = ldb(UH, v);
#pragma unroll
    for (int v = 0; v <= UL; ++v) hp[UH + 1 + v] = ldb(UL, v);
  }
  template <int G>
  static __device__ __forceinline__ void syrk36_store(double* tile, const f64x4* hp, const f64x4* acc, int lane, int valid,
                                                      bool damp, double lam, int ellipsoidal, double eps) {
    constexpr int UH = 4 + G, UL = 3 - G;
    auto st = [&](int u, int v, const f64x4& h, const f64x4& a) __attribute__((always_inline)) {
      const int r = 16 * u + (lane & 15), c0 = 16 * v + (lane >> 4);
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) {
        const int c = c0 + 4 * rho;
        double x = h[rho];
        if (u == v && r == c && damp) x = ellipsoidal ? x + (lam * x + eps) : x + lam;
        x -= a[rho];
        if (r >= valid || c >= valid) x = (r == c) ? 1.0 : 0.0;
        tile[tblk<double>(u >> 1, v >> 1) + (r & 31) * 34 + (c & 31)] = x;
      }
    };
#pragma unroll
    for (int v = 0; v <= UH; ++v) st(UH, v, hp[v], acc[v]);
#pragma unroll
    for (int v = 0; v <= UL; ++v) st(UL, v, hp[UH + 1 + v], acc[UH + 1 + v]);
  }
  static __device__ __forceinline__ void blk_zero(Blk& d) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) d.v[a][b][i] = 0.0;
  }
  static __device__ __forceinline__ void blk_sub(Blk& d, const Blk& a) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) d.v[p][q][i] -= a.v[p][q][i];
  }
  static __device__ __forceinline__ void blk_load(Blk& d, const double* blk, int lane) {
    const int rl = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) d.v[mh][nh][rho] = blk[(16 * nh + rl) * 34 + 16 * mh + kq + 4 * rho];
  }
  static __device__ __forceinline__ void blk_store(const Blk& d, double* blk, int lane, double sign) {
    const int rl = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) blk[(16 * nh + rl) * 34 + 16 * mh + kq + 4 * rho] = sign * d.v[mh][nh][rho];
  }
  static __device__ __forceinline__ void blk_mma(const double* Ablk, const double* Bblk, Blk& d, int lane, double asign) {
    const int rl = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const double a0 = asign * Ablk[rl * 34 + 4 * kk + kq], a1 = asign * Ablk[(16 + rl) * 34 + 4 * kk + kq];
      const double b0 = Bblk[rl * 34 + 4 * kk + kq], b1 = Bblk[(16 + rl) * 34 + 4 * kk + kq];
      d.v[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, d.v[0][0], 0, 0, 0);
      d.v[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, d.v[0][1], 0, 0, 0);
      d.v[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, d.v[1][0], 0, 0, 0);
      d.v[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, d.v[1][1], 0, 0, 0);
    }
  }

  // ---- register-resident blocks (see Engine<float>): Blk.v[kh][rowhalf][rho] = X[16 rowhalf + (lane&15)][16 kh + (lane>>4) + 4 rho],
  //      so MFMA (kh, rho) contracts the four consecutive columns 16 kh + 4 rho + {0..3} of both operands ----
  static __device__ __forceinline__ void blk_mma_rr(const Blk& A, const Blk& B, Blk& d, double asign) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) {
        const double a0 = asign * A.v[kh][0][rho], a1 = asign * A.v[kh][1][rho];
        const double b0 = B.v[kh][0][rho], b1 = B.v[kh][1][rho];
        d.v[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, d.v[0][0], 0, 0, 0);
        d.v[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, d.v[0][1], 0, 0, 0);
        d.v[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, d.v[1][0], 0, 0, 0);
        d.v[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, d.v[1][1], 0, 0, 0);
      }
  }
  static __device__ __forceinline__ void blk_neg(Blk& d) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) d.v[a][b][i] = -d.v[a][b][i];
  }
  static __device__ __forceinline__ void blk_load_global(Blk& d, const double* blk, const double* safe, int64_t ld, int rv, int cv,
                                                         bool diag, int lane) {
    const int rl = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const int r = 16 * nh + rl;
      const double* p = (r < rv ? blk + (int64_t)r * ld : safe) + kq;
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) {
          const int c = 16 * mh + kq + 4 * rho;
          const double x = p[16 * mh + 4 * rho];
          d.v[mh][nh][rho] = (r < rv && c < cv) ? x : ((diag && r == c) ? 1.0 : 0.0);
        }
    }
  }
  static __device__ __forceinline__ void blk_store_global(const Blk& d, double* blk, int64_t ld, int rv, int lane, double sign) {
    const int rl = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const int r = 16 * nh + rl;
      if (r < rv) {
        double* p = blk + (int64_t)r * ld + kq;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int rho = 0; rho < 4; ++rho) p[16 * mh + 4 * rho] = sign * d.v[mh][nh][rho];
      }
    }
  }
  static constexpr int NR = 2;
  static __device__ __forceinline__ int blk_row(int lane, int i) { return 16 * i + (lane & 15); }
  static __device__ __forceinline__ bool row_owner(int lane) { return lane < 16; }
  static __device__ __forceinline__ void blk_rowdot(const Blk& X, const double* vec, int lane, double (&out)[2]) {
    const int kq = lane >> 4;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      double s = 0.0;
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) s = __builtin_fma(X.v[mh][nh][rho], vec[16 * mh + kq + 4 * rho], s);
      s += __shfl_xor(s, 16);
      out[nh] = s + __shfl_xor(s, 32);
    }
  }
  template <int G>
  static __device__ __forceinline__ void syrk36_store_global(double* Lt, int64_t ld, const f64x4* hp, const f64x4* acc, int lane,
                                                             int valid, bool damp, double lam, int ellipsoidal, double eps) {
    constexpr int UH = 4 + G, UL = 3 - G;
    auto st = [&](int u, int v, const f64x4& h, const f64x4& a) __attribute__((always_inline)) {
      const int r = 16 * u + (lane & 15), c0 = 16 * v + (lane >> 4);
      if (r >= valid) return;
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) {
        const int c = c0 + 4 * rho;
        double x = h[rho];
        if (u == v && r == c && damp) x = ellipsoidal ? x + (lam * x + eps) : x + lam;
        Lt[(int64_t)r * ld + c] = x - a[rho];
      }
    };
#pragma unroll
    for (int v = 0; v <= UH; ++v) st(UH, v, hp[v], acc[v]);
#pragma unroll
    for (int v = 0; v <= UL; ++v) st(UL, v, hp[UH + 1 + v], acc[UH + 1 + v]);
  }
};

// Optional rider on the SYRK K-loop of chol_diag: the panel rows L_j,0:j pass through LDS anyway, so
// t[r] = sum_k L[row0+r][k] y[k] (the forward-substitution update) costs 16 VALU FMAs per thread and
// chunk in the shadow of the MFMAs.  Thread pair (2r, 2r+1) splits the chunk's k range in two.
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};

// ``after_issue`` runs right after the loads of the first k-chunk(s) have been ISSUED: whatever else a kernel wants in
// flight before its K-loop (H tile, solve panel) goes there, so that its latency overlaps the first chunk's instead of
// preceding it.
//
// ``ktiles`` (tile-sparse factorisation, thx_chol_factor_sparse): instead of the contiguous range [0, K) the loop visits the
// TILE-wide column blocks ktiles[0 .. K / TILE) -- the block columns in which BOTH operand row panels are structurally
// non-zero.  The list is wave uniform (scalar loads); skipping a block of exact zeros leaves every accumulator bit unchanged.
// SPLIT16 (fp64 SYRK): the chunk's columns [16 h, 16 h + 16) are staged as sub-chunk h at sA + h * 128 * LDT, row stride LDT.
// ``ksa`` / ``ksb`` (tile-packed factor): per K-list element the SLOT of the A / B operand tile; Arows / Brows then point at the
// problem's packed buffer, ``ld`` is TILE and ``packed_elems`` the buffer's extent (rows of a tile beyond the matrix are zero in
// the buffer itself, never written).
template <typename T, bool SAME, bool GEMV, int LDT, bool SPLIT16 = false, int NT = 256, int AHEAD_OVR = 0, int BROWS = TILE,
          typename Compute, typename Hook = NoHook>
__device__ __forceinline__ void kloop_f(const T* __restrict__ Arows, int validA, const T* __restrict__ Brows,
                                        int validB, int64_t ld, int K, T* sA, T* sB, int tid, const T* gemv_y,
                                        T* gemv_part, Compute&& compute, Hook&& after_issue = NoHook{},
                                        const int32_t* __restrict__ ktiles = nullptr, const int32_t* __restrict__ ksa = nullptr,
                                        const int32_t* __restrict__ ksb = nullptr, int64_t packed_elems = 0,
                                        bool gemv_compact = false) {
  // (gemv_compact: gemv_y holds the K-LIST's blocks of y back to back -- element kc * KB belongs to chunk kc -- instead of the
  //  whole vector: the level schedule's diagonal kernels, whose K-lists are short and scattered over all of y)
  using C = CT<T>;
  using V = typename C::V;
  constexpr int TPR = C::KB / C::VEC;   // threads per staged row (16 bytes each)
  static_assert(!GEMV || NT == 256, "the fused GEMV pairs the 256 threads with the 128 rows");
  constexpr int RPP = NT / TPR;         // rows per pass of the NT (256; the 8-wave fp64 off-diagonal kernel: 512) threads
  constexpr int NP = TILE / RPP;        // passes: fp32 4 x 32 rows, fp64 8 x 16 rows
  constexpr int NPB = BROWS / RPP;      // (operand B of the fp64 half-tile kernel: 64 rows -- the other passes are neither loaded nor staged)
  static_assert(BROWS % RPP == 0 && NPB >= 1 && NPB <= NP, "B rows: whole passes");
  const int lrow = tid / TPR, lc = tid % TPR;
  // element offset of this thread's 16-byte piece inside a staged row (set)
  const int scol = SPLIT16 ? ((lc * C::VEC) >> 4) * 128 * LDT + ((lc * C::VEC) & 15) : lc * C::VEC;
  // Operand rows through BUFFER loads: base pointer + extent live in a 4-SGPR resource, each lane contributes a 32-bit
  // byte offset, the k-chunk offset is a scalar -- no 64-bit per-row addresses in VGPRs (the fp64 kernels, two
  // workgroups per CU = 256 VGPRs, spilled them, and every reload put an s_waitcnt vmcnt(0) into the prefetch), and rows
  // outside the matrix are zeroed by the hardware bounds check (extent = valid rows) instead of v_cndmask / exec branches.
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const bool packed = ksa != nullptr;
  const int extA = packed ? (int)(packed_elems * (int64_t)sizeof(T)) : (int)((int64_t)validA * ld * (int64_t)sizeof(T));
  const int extB = packed ? extA : (int)((int64_t)(SAME ? validA : validB) * ld * (int64_t)sizeof(T));
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Arows), 0, extA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(SAME ? Arows : Brows), 0, extB, 0x00020000);
  unsigned voff[NP];
#pragma unroll
  for (int u = 0; u < NP; ++u) voff[u] = (unsigned)(((lrow + RPP * u) * (int)ld + lc * C::VEC) * (int)sizeof(T));
  // Register prefetch, AHEAD k-chunks deep: 128 bytes per staged row in flight.  fp32: one 32-column chunk (a second register
  // set measured no gain: 47.3-47.7 ms either way); fp64: two 16-column chunks -- a 16-column chunk is half the MFMA time of an
  // fp32 one, one chunk ahead does not cover the load latency (factor -2.9 %).
  constexpr int AHEAD = AHEAD_OVR ? AHEAD_OVR : (C::KB * (int)sizeof(T) <= 128 ? 2 : 1);   // (AHEAD_OVR: the fp64 half-tile kernel, 128 VGPRs)
  uint4 ra[AHEAD][NP], rb[AHEAD][NP];
  // (so.x / so.y: scalar byte offsets of the chunk inside the A / B operand buffers; the same unless the factor is tile-packed)
  auto gload = [&](uint4 (&xa)[NP], uint4 (&xb)[NP], int2 so) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(rsA, voff[u], so.x, 0);
      xa[u] = make_uint4(va.x, va.y, va.z, va.w);
      if (!SAME && u < NPB) {
        const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(rsB, voff[u], so.y, 0);
        xb[u] = make_uint4(vb.x, vb.y, vb.z, vb.w);
      }
    }
  };
  const int nk = K / C::KB;
  constexpr int CPT = TILE / C::KB;   // k-chunks per tile
  // first column of k-chunk kc
  // (the list is read through the constant address space: a SCALAR load.  As a plain global load the compiler issued a vector
  // load + s_waitcnt vmcnt(0) in front of every chunk's prefetch and wrapped each buffer load in a waterfall loop, since it
  // could not prove the offset wave uniform.)
  typedef const int32_t __attribute__((address_space(4))) * klist_t;
  const klist_t kl = (klist_t)(uintptr_t)ktiles;
  const klist_t ka = (klist_t)(uintptr_t)ksa, kb = (klist_t)(uintptr_t)ksb;
  auto kof = [&](int kc) __attribute__((always_inline)) -> int {
    const int k0 = kl ? kl[kc / CPT] * TILE + (kc % CPT) * C::KB : kc * C::KB;
    return __builtin_amdgcn_readfirstlane(k0);
  };
  // byte offsets of chunk kc in the two operand buffers
  auto sof = [&](int kc) __attribute__((always_inline)) -> int2 {
    if (!packed) {
      const int so = kof(kc) * (int)sizeof(T);
      return make_int2(so, so);
    }
    const int within = (kc % CPT) * C::KB;
    const int a = (ka[kc / CPT] * TILE * TILE + within) * (int)sizeof(T);
    const int bq = SAME ? a : (kb[kc / CPT] * TILE * TILE + within) * (int)sizeof(T);
    return make_int2(__builtin_amdgcn_readfirstlane(a), __builtin_amdgcn_readfirstlane(bq));
  };
  T gsum = T(0);
  // one k-chunk: registers -> LDS, refill the registers with the chunk AHEAD steps on, MFMAs on the staged chunk
  // (a static s_setprio per hardware wave slot, to push the two co-resident workgroups out of lockstep, measured no gain)
  auto step = [&](uint4 (&xa)[NP], uint4 (&xb)[NP], int kc) __attribute__((always_inline)) {
    // column of the chunk to prefetch: the K-list entry is fetched here so that its (scalar) load completes under the staging
    const int2 knext = kc + AHEAD < nk ? sof(kc + AHEAD) : make_int2(0, 0);
#ifdef THX_EXP_NOSTAGE  // timing experiment: no register -> LDS staging and only one barrier per chunk (garbage results)
    if (kc == 0) {
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int row = lrow + RPP * u;
        *reinterpret_cast<uint4*>(sA + row * LDT + scol) = xa[u];
        if (!SAME) *reinterpret_cast<uint4*>(sB + row * LDT + scol) = xb[u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        asm volatile("" ::"v"(xa[u].x));  // the loads must still complete
        if (!SAME) asm volatile("" ::"v"(xb[u].x));
      }
    }
    __syncthreads();
#else
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int row = lrow + RPP * u;
      *reinterpret_cast<uint4*>(sA + row * LDT + scol) = xa[u];
      if (!SAME && u < NPB) *reinterpret_cast<uint4*>(sB + row * LDT + scol) = xb[u];
    }
    __syncthreads();
#endif
#ifdef THX_EXP_NOLOAD
    if (kc == 0 && kc + AHEAD < nk) gload(xa, xb, knext);  // timing experiment: operands are not streamed
#else
    if (kc + AHEAD < nk) gload(xa, xb, knext);
#endif
    if constexpr (GEMV) {
      if (gemv_y) {
        constexpr int HALF = C::KB / 2;
        // (thread pair (2r, 2r+1): the two 16-column halves of row r -- with SPLIT16 the two sub-chunks)
        const V* rp = reinterpret_cast<const V*>(sA + (tid >> 1) * LDT + (SPLIT16 ? (tid & 1) * 128 * LDT : (tid & 1) * HALF));
        const V* yp = reinterpret_cast<const V*>(gemv_y + (gemv_compact ? kc * C::KB : kof(kc)) + (tid & 1) * HALF);
#pragma unroll
        for (int i = 0; i < HALF / C::VEC; ++i) {
          const V a = rp[i], yv = yp[i];
          if constexpr (sizeof(T) == 4) {
            gsum += a.x * yv.x; gsum += a.y * yv.y; gsum += a.z * yv.z; gsum += a.w * yv.w;
          } else {
            gsum += a.x * yv.x; gsum += a.y * yv.y;
          }
        }
      }
    }
#ifdef THX_EXP_NOMFMA
    if (K < 0) compute();  // timing experiment: no MFMAs (garbage results)
#else
    compute();  // MFMAs on the staged chunk (sA / sB)
#endif
  };
#pragma unroll
  for (int a = 0; a < AHEAD; ++a)
    if (a < nk) gload(ra[a], rb[a], sof(a));
  after_issue();
  for (int kc = 0; kc < nk; kc += AHEAD) {
    step(ra[0], rb[0], kc);
    if constexpr (AHEAD == 2) {
      if (kc + 1 < nk) step(ra[1], rb[1], kc + 1);
    }
  }
  if constexpr (GEMV) {
    if (gemv_part) *gemv_part = gsum;
  }
}

template <typename T, bool SAME, bool GEMV = false, typename Hook = NoHook>
__device__ __forceinline__ void kloop(const T* __restrict__ Arows, int validA, const T* __restrict__ Brows,
                                      int validB, int64_t ld, int K, T* sA, T* sB,
                                      typename Engine<T>::Acc& acc, int tid, const T* gemv_y = nullptr,
                                      T* gemv_part = nullptr, Hook&& after_issue = NoHook{},
                                      const int32_t* __restrict__ ktiles = nullptr, const int32_t* __restrict__ ksa = nullptr,
                                      const int32_t* __restrict__ ksb = nullptr, int64_t packed_elems = 0) {
  const int wave = tid >> 6, lane = tid & 63;
  kloop_f<T, SAME, GEMV, CT<T>::LDT>(Arows, validA, Brows, validB, ld, K, sA, sB, tid, gemv_y, gemv_part, [&]() __attribute__((always_inline)) {
    Engine<T>::chunk(sA, (SAME ? sA : sB) + 32 * wave * CT<T>::LDT, acc, lane);
  }, after_issue, ktiles, ksa, ksb, packed_elems);
}


// ------------------------------------------------------------------------------------------------
// blocked substitutions with a panel M in LDS (diag sub-blocks W_ss = L_ss^-1, below: -L_st)
// executed by wave 0 (64 lanes: lane = (row-in-block, half of the column range)); the caller
// brackets them with __syncthreads().  vec holds the right-hand side on entry, the solution on exit.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_sum(float x) { return x + __shfl_xor(x, 32); }
__device__ __forceinline__ double half_sum(double x) { return x + __shfl_xor(x, 32); }

// y = L_jj^-1 v :  for s: u_s = v_s + sum_{c < 32s} M[r][c] y[c] ;  y_s = W_ss u_s
template <typename T>
__device__ __forceinline__ void panel_forward(const T* M, T* vec, T* ubuf, int lane) {
  using C = CT<T>;
  const int rl = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int sb = 0; sb < 4; ++sb) {
    const T* row = M + (32 * sb + rl) * C::LDM;
    T u = T(0);
    // columns [0, 32 sb) split between the two lane halves in 16-column slabs
    for (int c = 16 * hf; c < 32 * sb; c += 32)
#pragma unroll
      for (int i = 0; i < 16; ++i) u += row[c + i] * vec[c + i];
    u = half_sum(u) + vec[32 * sb + rl];
    if (hf == 0) ubuf[rl] = u;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes are visible to its reads
    __builtin_amdgcn_wave_barrier();
    T yv = T(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) yv += row[32 * sb + 16 * hf + i] * ubuf[16 * hf + i];
    yv = half_sum(yv);
    __builtin_amdgcn_wave_barrier();
    if (hf == 0) vec[32 * sb + rl] = yv;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

// x = L_jj^-T z :  for t = 3..0: a_t = z_t + sum_{r >= 32(t+1)} M[r][c] x[r] ;  x_t = W_tt^T a_t
template <typename T>
__device__ __forceinline__ void panel_backward(const T* M, T* vec, T* ubuf, int lane) {
  using C = CT<T>;
  const int cl = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int tb = 3; tb >= 0; --tb) {
    const T* col = M + 32 * tb + cl;
    T a = T(0);
    for (int r = 32 * (tb + 1) + 16 * hf; r < TILE; r += 32)
#pragma unroll
      for (int i = 0; i < 16; ++i) a += col[(r + i) * C::LDM] * vec[r + i];
    a = half_sum(a) + vec[32 * tb + cl];
    if (hf == 0) ubuf[cl] = a;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    T xv = T(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) xv += col[(32 * tb + 16 * hf + i) * C::LDM] * ubuf[16 * hf + i];
    xv = half_sum(xv);
    __builtin_amdgcn_wave_barrier();
    if (hf == 0) vec[32 * tb + cl] = xv;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// 32x32 in-wave kernels of the diagonal-tile factorisation.  Straight-line code is kept SMALL and is
// re-used by a run-time loop over the four sub-blocks: a fully unrolled 128-step factorisation is
// ~120 KB of instructions, streams through the 64 KB instruction cache once per workgroup and runs at
// L2 instruction-fetch latency (measured: 2700 cycles per 120-instruction step).
// ------------------------------------------------------------------------------------------------
// lane r (= lane & 31) holds row r: a[c] = S[r][c].  On exit a[c] = L[r][c] for c <= r (columns above the
// diagonal are garbage).  Broadcasts go through SGPRs (v_readlane), no LDS round trips.  Returns the
// 1-based index of the first non-positive pivot (0 = positive definite).
template <typename T, int N>
__device__ __forceinline__ int potrf_reg(T (&a)[N]) {
  int bad = 0;
  static_for<N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int c = decltype(ic)::value;
    T d = bcast(a[c], c);
    if (!(d > T(0))) {
      if (bad == 0) bad = c + 1;
      d = T(1);
    }
    const T isq = t_rsqrt(d);
    a[c] *= isq;  // L[r][c]
    static_for<N - 1 - c>([&](auto iq) __attribute__((always_inline)) {
      constexpr int q = c + 1 + decltype(iq)::value;
      a[q] -= a[c] * bcast(a[c], q);  // S[r][q] -= L[r][c] L[q][c]
    });
  });
  return bad;
}
template <typename T>
__device__ __forceinline__ int potrf32(T (&a)[32]) {
  return potrf_reg<T, 32>(a);
}

// W = L^-1 for a 32x32 triangle stored row-major in LDS (row stride LDM): lane j (= lane & 31) computes column j of
// W by forward substitution on e_j.  L[i][k] is wave uniform: it is read with broadcast ds_read_b128 (4 values per
// instruction, into VGPRs) -- a v_readlane per value does not scale: hipcc hoists all 496 of them, runs out of SGPRs
// and spills through v_writelane/v_readlane pairs (measured 30 cycles per term).  Row i+1 is loaded while row i is
// being consumed.  Accumulation in A: fp64 for the fp64 path, float for fp32 -- W then carries the backward error of
// an fp32 TRSM (columnwise ~32 eps |L|), which is what the sub-block solve it replaces would have.
template <typename T, typename A, int LDM, int N>
__device__ __forceinline__ void inv_tri(const T* Lss, A (&w)[N], int lane) {
  constexpr int VEC = 16 / sizeof(T);
  using V = std::conditional_t<sizeof(T) == 4, float4, double2>;
  const int j = lane & (N - 1);
  V cur[N / VEC], nxt[N / VEC];
  cur[0] = *reinterpret_cast<const V*>(Lss);
  static_for<N>([&](auto ii) __attribute__((always_inline)) {
    constexpr int i = decltype(ii)::value;
    if constexpr (i + 1 < N) {  // prefetch row i+1: entries 0 .. i+1
#pragma unroll
      for (int q = 0; q <= (i + 1) / VEC; ++q) nxt[q] = *reinterpret_cast<const V*>(Lss + (i + 1) * LDM + VEC * q);
    }
    auto at = [&](int k) __attribute__((always_inline)) -> A {
      if constexpr (sizeof(T) == 4) {
        const float4 v = cur[k >> 2];
        return (A)((k & 3) == 0 ? v.x : (k & 3) == 1 ? v.y : (k & 3) == 2 ? v.z : v.w);
      } else {
        const double2 v = cur[k >> 1];
        return (A)((k & 1) ? v.y : v.x);
      }
    };
    const A lii = at(i);
    A r = (A)(1.0f / (float)lii);
    r = r * (A(2) - lii * r);
    r = r * (A(2) - lii * r);
    A s0 = (j == i) ? A(1) : A(0), s1 = A(0);
    static_for<i>([&](auto kk) __attribute__((always_inline)) {
      constexpr int k = decltype(kk)::value;
      if constexpr (k & 1) s1 -= at(k) * w[k];
      else s0 -= at(k) * w[k];
    });
    w[i] = (s0 + s1) * r;
    if constexpr (i + 1 < N) {
#pragma unroll
      for (int q = 0; q <= (i + 1) / VEC; ++q) cur[q] = nxt[q];
    }
    // keep the rows in order: without this the compiler issues every LDS read first and the FMA chains sink below
    // them (hundreds of spilled registers)
    asm volatile("" : "+v"(w[i]) : : "memory");
  });
}
template <typename T, typename A, int LDM>
__device__ __forceinline__ void inv32(const T* Lss, A (&w)[32], int lane) {
  inv_tri<T, A, LDM, 32>(Lss, w, lane);
}

// ------------------------------------------------------------------------------------------------
// (The scheme of rounds 2-6, kept for A/B under -DTHX_POTRF_BLOCKED; the default is potrf_inv32_lanes below: 7 us less per tile,
//  profiles/r6/ah_.)
// One 32x32 diagonal sub-block, blocked 16 + 16 (executed by ONE wave): the in-register factorisation and the
// substitution for the inverse cost ~N^2 dependent readlane/FMA steps each, so halving N and doing the coupling
// with 16x16x16 MFMA products on the LDS block halves the serial chain of chol_diag:
//   L00 = chol(S00), W00 = L00^-1, L10 = S10 W00^T, S11 -= L10 L10^T, L11 = chol(S11), W11 = L11^-1,
//   W10 = -W11 (L10 W00);   on exit the block holds W_ss = [[W00, 0], [W10, W11]] and L_ss has gone to global memory.
// 16x16 MFMA convention (both dtypes):  acc(n, m) += sum_k Aop[m][k] * B[n][k],  n = lane & 15 (row of the B block),
// m = m_of(lane, i) for accumulator element i (fp32: 4 (lane >> 4) + i; fp64: (lane >> 4) + 4 i); TA reads A transposed.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct Mma16;
template <>
struct Mma16<float> {
  using Acc = f32x4;
  static __device__ __forceinline__ int m_of(int lane, int i) { return 4 * (lane >> 4) + i; }
  template <bool TA>
  static __device__ __forceinline__ void mma(const float* A, const float* B, Acc& acc, int lane, float asign) {
    constexpr int LDB = CT<float>::LDB;
    const int x = lane & 15, kq = lane >> 4;
    const float4 fb = *reinterpret_cast<const float4*>(B + x * LDB + 4 * kq);
    float fa[4];
    if constexpr (TA) {
#pragma unroll
      for (int e = 0; e < 4; ++e) fa[e] = A[(4 * kq + e) * LDB + x];
    } else {
      const float4 v = *reinterpret_cast<const float4*>(A + x * LDB + 4 * kq);
      fa[0] = v.x; fa[1] = v.y; fa[2] = v.z; fa[3] = v.w;
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(asign * fa[0], fb.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(asign * fa[1], fb.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(asign * fa[2], fb.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(asign * fa[3], fb.w, acc, 0, 0, 0);
  }
};
template <>
struct Mma16<double> {
  using Acc = f64x4;
  static __device__ __forceinline__ int m_of(int lane, int i) { return (lane >> 4) + 4 * i; }
  template <bool TA>
  static __device__ __forceinline__ void mma(const double* A, const double* B, Acc& acc, int lane, double asign) {
    constexpr int LDB = CT<double>::LDB;
    const int x = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double fa = TA ? A[(4 * e + kq) * LDB + x] : A[x * LDB + 4 * e + kq];
      const double fb = B[x * LDB + 4 * e + kq];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(asign * fa, fb, acc, 0, 0, 0);
    }
  }
};

// the wave's own LDS writes become visible to its reads
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
}

// Dss: the 32 x LDB block in LDS (S_ss on entry, W_ss on exit); Lg: global address of L's element (first row of the
// sub-block, first column of the sub-block), rows_valid = number of the sub-block's rows inside the matrix.
// Returns the 1-based index (within the sub-block) of the first non-positive pivot, 0 if none.
template <typename T>
__device__ __forceinline__ int potrf_inv32_blocked(T* Dss, T* Lg, int64_t ld, int rows_valid, int lane) {
  using C = CT<T>;
  using V = typename C::V;
  using M = Mma16<T>;
  using WA = std::conditional_t<sizeof(T) == 8, double, float>;
  constexpr int LDB = C::LDB;
  T* Q00 = Dss;
  T* Q01 = Dss + 16;
  T* Q10 = Dss + 16 * LDB;
  T* Q11 = Dss + 16 * LDB + 16;
  const int r = lane & 15;
  int bad = 0;
  // row r of a 16x16 LDS quadrant -> registers
  auto load_row = [&](const T* Q, T (&a)[16]) __attribute__((always_inline)) {
    const V* rp = reinterpret_cast<const V*>(Q + r * LDB);
#pragma unroll
    for (int q = 0; q < 16 / C::VEC; ++q) {
      const V v = rp[q];
      if constexpr (sizeof(T) == 4) {
        a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
      } else {
        a[2 * q] = v.x; a[2 * q + 1] = v.y;
      }
    }
  };
  // lower-triangular rows (zeros above the diagonal) -> the LDS quadrant and -> global memory (rows inside the matrix)
  auto store_tri = [&](T* Q, T* G, int row_in_block, const T (&a)[16]) __attribute__((always_inline)) {
    if (lane < 16) {
      V* rp = reinterpret_cast<V*>(Q + r * LDB);
      V* gp = reinterpret_cast<V*>(G + (int64_t)r * ld);
      const bool g = row_in_block + r < rows_valid;
#pragma unroll
      for (int q = 0; q < 16 / C::VEC; ++q) {
        V v;
        if constexpr (sizeof(T) == 4)
          v = make_float4(4 * q <= r ? a[4 * q] : 0.f, 4 * q + 1 <= r ? a[4 * q + 1] : 0.f,
                          4 * q + 2 <= r ? a[4 * q + 2] : 0.f, 4 * q + 3 <= r ? a[4 * q + 3] : 0.f);
        else
          v = make_double2(2 * q <= r ? a[2 * q] : 0.0, 2 * q + 1 <= r ? a[2 * q + 1] : 0.0);
        rp[q] = v;
        if (g) gp[q] = v;
      }
    }
  };
  // column j = r of W (w[i] = W[i][j], zero above the diagonal) -> row-major into the quadrant
  auto store_cols = [&](T* Q, const WA (&w)[16]) __attribute__((always_inline)) {
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) Q[i * LDB + r] = (T)w[i];
    }
  };
  auto acc_zero = [&](typename M::Acc& acc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = T(0);
  };
  auto acc_store = [&](const typename M::Acc& acc, T* Q) __attribute__((always_inline)) {
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(Q + r * LDB + 4 * (lane >> 4)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) Q[r * LDB + M::m_of(lane, i)] = acc[i];
    }
  };

  T a[16];
  WA w[16];
  typename M::Acc acc;
  // ---- L00, W00 ----
  load_row(Q00, a);
  bad = potrf_reg<T, 16>(a);
  store_tri(Q00, Lg, 0, a);
  wave_lds_fence();
  inv_tri<T, WA, LDB, 16>(Q00, w, lane);
  __builtin_amdgcn_wave_barrier();
  store_cols(Q00, w);
  wave_lds_fence();
  // ---- L10 = S10 W00^T ----
  acc_zero(acc);
  M::template mma<false>(Q00, Q10, acc, lane, T(1));
  acc_store(acc, Q10);
  if (16 + r < rows_valid) {  // L10 -> global
    T* gp = Lg + (int64_t)(16 + r) * ld;
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(gp + 4 * (lane >> 4)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) gp[M::m_of(lane, i)] = acc[i];
    }
  }
  wave_lds_fence();
  // ---- S11 -= L10 L10^T ----
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = Q11[r * LDB + M::m_of(lane, i)];
  M::template mma<false>(Q10, Q10, acc, lane, T(-1));
  acc_store(acc, Q11);
  wave_lds_fence();
  // ---- L11, W11 ----
  load_row(Q11, a);
  {
    const int bad1 = potrf_reg<T, 16>(a);
    if (bad == 0 && bad1 != 0) bad = 16 + bad1;
  }
  store_tri(Q11, Lg + 16 * ld + 16, 16, a);
  wave_lds_fence();
  inv_tri<T, WA, LDB, 16>(Q11, w, lane);
  __builtin_amdgcn_wave_barrier();
  store_cols(Q11, w);
  wave_lds_fence();
  // ---- W10 = -W11 (L10 W00): T1 = L10 W00 through the (otherwise zero) upper-right quadrant ----
  acc_zero(acc);
  M::template mma<true>(Q00, Q10, acc, lane, T(1));   // T1(n = row of L10, m = c) = sum_k W00[k][c] L10[n][k]
  acc_store(acc, Q01);
  wave_lds_fence();
  acc_zero(acc);
  M::template mma<true>(Q01, Q11, acc, lane, T(-1));  // W10(n, c) = -sum_k T1[k][c] W11[n][k]
  acc_store(acc, Q10);
  if (lane < 16) {  // upper-right quadrant back to zero: W_ss is used as a full 32x32 operand
    V* rp = reinterpret_cast<V*>(Q01 + r * LDB);
#pragma unroll
    for (int q = 0; q < 16 / C::VEC; ++q) {
      if constexpr (sizeof(T) == 4) rp[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      else rp[q] = make_double2(0.0, 0.0);
    }
  }
  return bad;
}

// The same contract as potrf_inv32_blocked with the inverse for FREE: lanes 0..31 hold the rows of S_ss, lanes 32..63 the rows
// of the identity, and the factorisation's column operations (column c scaled by 1/sqrt(pivot), column q -= column c * L[q][c])
// run over all 64 lanes in the same instructions.  S -> L = S U with U = L^-T, so the identity becomes U: lane 32 + r ends with
// a[q] = U[r][q] = W[q][r], exact zeros for q < r -- column r of W = L^-1, what inv_tri produced with a second N^2 / 2 chain of
// dependent FMAs, two LDS round trips and five 16 x 16 MFMA couplings around it.  One wave issues in order, so the chain's cost
// is its instruction count: 32 steps of (pivot broadcast, rsqrt, scale) + 496 (readlane, fma) pairs.
template <typename T>
__device__ __forceinline__ int potrf_inv32_lanes(T* Dss, T* Lg, int64_t ld, int rows_valid, int lane) {
  using C = CT<T>;
  using V = typename C::V;
  constexpr int LDB = C::LDB;
  const int r = lane & 31;
  const bool upper = lane >= 32;
  T a[32];
  {
    const V* rp = reinterpret_cast<const V*>(Dss + r * LDB);
#pragma unroll
    for (int q = 0; q < 32 / C::VEC; ++q) {
      const V v = rp[q];
      if constexpr (sizeof(T) == 4) {
        a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
      } else {
        a[2 * q] = v.x; a[2 * q + 1] = v.y;
      }
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) a[q] = upper ? (q == r ? T(1) : T(0)) : a[q];
  }
  __builtin_amdgcn_wave_barrier();   // (every lane has read its row before column r of W overwrites the block)
  const int bad = potrf_reg<T, 32>(a);
  if (!upper) {   // L_ss -> global memory: one 32-element row per lane, zeros above the diagonal
    if (r < rows_valid) {
      V* gp = reinterpret_cast<V*>(Lg + (int64_t)r * ld);
#pragma unroll
      for (int q = 0; q < 32 / C::VEC; ++q) {
        if constexpr (sizeof(T) == 4)
          gp[q] = make_float4(4 * q <= r ? a[4 * q] : 0.f, 4 * q + 1 <= r ? a[4 * q + 1] : 0.f, 4 * q + 2 <= r ? a[4 * q + 2] : 0.f,
                              4 * q + 3 <= r ? a[4 * q + 3] : 0.f);
        else
          gp[q] = make_double2(2 * q <= r ? a[2 * q] : 0.0, 2 * q + 1 <= r ? a[2 * q + 1] : 0.0);
      }
    }
  } else {        // W_ss -> the LDS block, row-major: W[q][r] (zero above the diagonal by construction)
#pragma unroll
    for (int q = 0; q < 32; ++q) Dss[q * LDB + r] = a[q];
  }
  return bad;
}
template <typename T>
__device__ __forceinline__ int potrf_inv32(T* Dss, T* Lg, int64_t ld, int rows_valid, int lane) {
#ifdef THX_POTRF_BLOCKED
  return potrf_inv32_blocked<T>(Dss, Lg, ld, rows_valid, lane);
#else
  return potrf_inv32_lanes<T>(Dss, Lg, ld, rows_valid, lane);
#endif
}

// ------------------------------------------------------------------------------------------------
// chol_diag: SYRK + blocked Cholesky of the 128x128 diagonal tile of block column j, panel M_j,
// fused forward substitution
// ------------------------------------------------------------------------------------------------
// device view of thx_tile_pattern (include/theseus_hip.h): which 128x128 tiles of L are structurally non-zero
struct TilePat {
  const int32_t* col_ptr;    // (ntiles + 1) entries of block column j: [col_ptr[j], col_ptr[j + 1])
  const int32_t* col_row;    // row tile of every entry (ascending within a column)
  const int32_t* tile_kptr;  // (entries + 1) K-list of entry e ...
  const int32_t* tile_k;     // ... block columns k < j in which L_ik and L_jk are both non-zero
  const int32_t* diag_kptr;  // (ntiles + 1) K-list of diagonal tile j ...
  const int32_t* diag_k;     // ... block columns k < j with L_jk non-zero
  // TILE-PACKED factor (nslots > 0): L is (B, nslots, TILE, TILE) -- only the tiles of the pattern exist: slot j = diagonal tile
  // j, slot ntiles + e = off-diagonal entry e -- and every K-list element carries the slots of its two operand tiles
  const int32_t* tile_sa;    // (per tile_k element) slot of L_jk
  const int32_t* tile_sb;    //                      slot of L_ik
  const int32_t* diag_s;     // (per diag_k element) slot of L_jk
  int32_t nslots;            // 0: L is the dense (B, ld, ld) frame
  int32_t lpt;               // chol_offdiag's block -> (problem, entry) map: 1 = the ENTRY is the slow index (longest K-lists first)
  // LEVEL schedule (thx_chol_factor_levels): one launch covers every block column of an elimination-tree level
  const int32_t* ent_col;    // (entries) block column of entry e -- the launch's entries then are [i_first, i_first + nrow_tiles)
  const int32_t* tile_valid; // (ntiles) rows / columns of tile j inside the matrix, the rest is identity padding (per-tile padding:
                             // no variable straddles a tile boundary); nullptr: min(TILE, n - j * TILE)
  // RIGHT-LOOKING schedule of small dense batches (factor_impl: "rl"): 0 = the left-looking kernels as they are; 1 = no K-loop
  // (the tile read from the H argument already carries every earlier column's update: chol_diag = the tile factorisation alone,
  // chol_offdiag = the substitution alone); 2 + jc = chol_offdiag as the TRAILING UPDATE of block column jc: workgroup slot t ->
  // tile (i, k), jc < k <= i, receives  A_ik - L_i,jc L_k,jc^T  (no substitution), written to the L frame
  int32_t rl;
  // right-looking schedule with a right-hand side: the vector being forward-substituted, (B, rl_ldv) -- g on entry; chol_diag
  // turns block j into y_j in place, every substitution tile (i, j) then takes  L_ij y_j  off block i (chol_offdiag, rl == 1)
  void* rl_y;
  int64_t rl_ldv;
  // right-looking schedule, two launches per block column (rl == 1): the tile takes the PREVIOUS block column's update itself -- a
  // K-loop over the one tile of column j - 1, the left-looking kernels' own path -- instead of finding it applied: chol_diag(j)
  // and the substitution tiles (i, j) of chol_offdiag
  int32_t rl_la;
  // ... and ONE chol_offdiag launch per block column (rl == 1, rl_la == 1, rl_nsub > 0): slots [0, rl_nsub) = the substitution
  // tiles (i, jarg), each taking column jarg - 1's update of itself first (a K-loop over the one tile of that column); the other
  // slots = column jarg - 1's trailing update of the tiles (i, k), jarg < k <= i -- which nobody needs before column jarg + 1
  int32_t rl_nsub;
};

// rows (= columns) of diagonal tile j that belong to the matrix
__device__ __forceinline__ int tile_rows(const TilePat& pat, int n, int j) {
  return pat.tile_valid ? pat.tile_valid[j] : min(TILE, n - j * TILE);
}

// where a kernel finds / puts the tiles of L: the dense frame (row stride ld) or the tile-packed buffer (row stride TILE)
struct LFrame {
  int64_t pstride;   // elements per problem
  int64_t ld;        // row stride of a tile
  bool packed;
  __device__ __forceinline__ int64_t tile(int ti, int tj, int slot) const {   // element offset of tile (ti, tj) inside a problem
    return packed ? (int64_t)slot * TILE * TILE : (int64_t)ti * TILE * ld + (int64_t)tj * TILE;
  }
};
__device__ __forceinline__ LFrame lframe(const TilePat& pat, int64_t ld) {
  const bool packed = pat.nslots > 0;
  return LFrame{packed ? (int64_t)pat.nslots * TILE * TILE : ld * ld, packed ? (int64_t)TILE : ld, packed};
}

// Fused forward substitution through a finished panel column (sub-block column sb of the diagonal tile), on blocks in the
// register layout of Engine::Blk:  y_s = W_ss u_s ;  u_u += (-L_us) y_s  for the sub-blocks below.  ONE implementation for both
// schedules of the diagonal phase (chol_diag_kernel loads the blocks from its LDS tile, chol_potrf_kernel has them in
// registers): the same sums in the same order, so y -- and with it every LM trajectory -- does not depend on which schedule a
// batch size selects (tests/test_gpu_full_size.py: any slice of a batch solved alone is bit-identical).
template <typename T>
__device__ __forceinline__ void fwd_diag_block(const typename Engine<T>::Blk& W, T* vvec, int sb, int lane) {
  using E = Engine<T>;
  T y[E::NR];
  E::blk_rowdot(W, vvec + 32 * sb, lane, y);
  wave_lds_fence();   // every lane has read u_s
  if (E::row_owner(lane)) {
#pragma unroll
    for (int i = 0; i < E::NR; ++i) vvec[32 * sb + E::blk_row(lane, i)] = y[i];
  }
  wave_lds_fence();
}
template <typename T>
__device__ __forceinline__ void fwd_below_block(const typename Engine<T>::Blk& X, T* vvec, int sb, int u, int lane) {
  using E = Engine<T>;
  T part[E::NR];
  E::blk_rowdot(X, vvec + 32 * sb, lane, part);
  if (E::row_owner(lane)) {
#pragma unroll
    for (int i = 0; i < E::NR; ++i) vvec[32 * u + E::blk_row(lane, i)] += part[i];
  }
}

// device view of a block-compact Hessian (include/theseus_hip.h: thx_hblock_layout + the value buffer); blocks == nullptr:
// H is the dense frame
struct HBlk {
  const void* blocks;
  int64_t bstride;
  int bd;
  const int32_t* tile_ptr;
  const int32_t* piece_blk;
  const int32_t* piece_rc;
  const int32_t* diag_blk;   // (nvars) block id of variable v's diagonal block (the right-looking schedule's damping pass; may be null)
  int max_tile_pieces;       // thx_hblock_layout.max_tile_pieces (host side: picks the off-diagonal kernels' HB mode); 0: unknown
};

// The pieces of lower tile (ti, tj) of problem b -- f(r, c, value), (r, c) relative to the tile origin and inside the tile; the
// blocks of a tile are one contiguous run of the list (straddlers from the neighbours aside): coalesced reads -- fetched EARLY:
// the first NPRE x 256 elements of the tile go global -> registers in the kernel's prologue (two
// dependent loads each -- table, then value: ~2 us if left to the epilogue, measured as +1.3 ms per factorisation), the K-loop
// hides them; ``foreach`` then replays them from registers (and walks whatever is beyond NPRE x 256 from memory).
template <typename T, int NPRE, int NT = 256>   // (NT: threads of the workgroup)
struct HBPre {
  static constexpr int CAP = NT * NPRE;   // elements the registers hold
  T v[NPRE];
  int rc[NPRE];   // (r << 8) | c inside the tile, -1: nothing
  int p0, cnt;
  int wmeta;      // lane l of every wave: piece_rc of the tile's piece l (hb_scatter: readlane)
  __device__ __forceinline__ void empty() {   // (timing experiments)
    p0 = cnt = wmeta = 0;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      rc[k] = -1;
      v[k] = T(0);
    }
  }
  // the tile's elements, in list order, -> LDS (element idx of the tile's run at list[idx]; the caller publishes them with a barrier)
  __device__ __forceinline__ void to_list(T* list, int tid) const {
#pragma unroll
    for (int k = 0; k < NPRE; ++k)
      if (tid + NT * k < cnt) list[tid + NT * k] = v[k];   // (k < NPRE: what the registers hold)
  }
  // BRANCH-FREE (round 5): the loads of all NPRE elements are independent of each other -- with an ``if (idx < cnt)`` around each
  // element hipcc emitted  table load, s_waitcnt vmcnt(0), table load, s_waitcnt vmcnt(0), value load  once per element, i.e.
  // 2 NPRE exposed round trips at the head of every workgroup (6 per off-diagonal tile, 14 per SYRK workgroup).  Now: both tables
  // of all elements in one batch, one wait, all values in one batch whose wait is the first use (after the K-loop).  A lane
  // without an element reads element 0 of the tile's first piece and discards it.
  // (Round 6 measured the chain in TWO PHASES -- tile_ptr -> piece_rc / piece_blk at the kernel's very top, the values behind the
  //  first k-chunk's loads: no gain in fp64, 0.4 of 43.5 ms SLOWER in fp32, profiles/r6/ab_.  load() keeps both in one place; the
  //  split stays as two functions.)
  int tw[NPRE], tblk[NPRE];   // (live between the phases only)
  __device__ __forceinline__ void load_tables(const HBlk& hb, int ti, int tj, int tid) {
    const int bd = hb.bd, bb = bd * bd, t = ti * (ti + 1) / 2 + tj;
    p0 = hb.tile_ptr[t];
    cnt = (hb.tile_ptr[t + 1] - p0) * bb;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      rc[k] = -1;
      v[k] = T(0);
      tw[k] = tblk[k] = 0;
    }
    wmeta = 0;
    if (cnt <= 0) return;   // (workgroup uniform; p0 may be the END of the piece list)
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int idx = tid + NT * k;
      const int pc = p0 + (idx < cnt ? idx : 0) / bb;
      tw[k] = hb.piece_rc[pc];
      tblk[k] = hb.piece_blk[pc];
    }
    wmeta = hb.piece_rc[p0 + min(tid & 63, cnt / bb - 1)];
  }
  // BRANCH-FREE (round 5): the loads of all NPRE elements are independent of each other -- all values in one batch whose wait is
  // the first use (after the K-loop).  A lane without an element reads element 0 of the tile's first piece and discards it.
  __device__ __forceinline__ void load_values(const HBlk& hb, int b, int tid) {
    if (cnt <= 0) return;
    const T* base = static_cast<const T*>(hb.blocks) + (int64_t)b * hb.bstride;
    const int bd = hb.bd, bb = bd * bd;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int idx = tid + NT * k;
      const bool ok = idx < cnt;
      const int e = (ok ? idx : 0) % bb;
      const int r = (int)(short)(tw[k] >> 16) + e / bd, c = (int)(short)(tw[k] & 0xffff) + e % bd;
      const T val = base[(int64_t)tblk[k] * bb + e];
      if (ok && r >= 0 && r < TILE && c >= 0 && c < TILE) {
        rc[k] = (r << 8) | c;
        v[k] = val;
      }
    }
  }
  __device__ __forceinline__ void load(const HBlk& hb, int b, int ti, int tj, int tid) {
    load_tables(hb, ti, tj, tid);
    load_values(hb, b, tid);
  }
  template <typename F>
  __device__ __forceinline__ void foreach(const HBlk& hb, int b, int tid, F&& f) const {
#pragma unroll
    for (int k = 0; k < NPRE; ++k)
      if (rc[k] >= 0) f(rc[k] >> 8, rc[k] & 255, v[k]);
    if (cnt > NT * NPRE) {   // a tile with more pieces than the registers hold
      const T* base = static_cast<const T*>(hb.blocks) + (int64_t)b * hb.bstride;
      const int bd = hb.bd, bb = bd * bd;
      if (bd == 6) {
        // ONE PIECE PER THREAD (6 x 6 blocks: the reduced camera system of a bundle adjustment fills a tile with up to 21 x 21
        // of them -- 15876 elements; element by element that was 62 rounds of  table, table, value  per thread): a block is 36
        // contiguous values, read as three batches of twelve (16-byte vectors), its table entries once.  The piece that holds
        // element 256 NPRE of the tile's run is split with the register part above.
        constexpr int VEC = 16 / sizeof(T), NV = 12 / VEC;
        typedef T TV __attribute__((ext_vector_type(VEC)));
        const int pb = (NT * NPRE) / 36, eb = (NT * NPRE) % 36, np = cnt / 36;
        for (int q = pb + tid; q < np; q += NT) {
          const int w = hb.piece_rc[p0 + q];
          const int r0 = (int)(short)(w >> 16), c0 = (int)(short)(w & 0xffff);
          const TV* src = reinterpret_cast<const TV*>(base + (int64_t)hb.piece_blk[p0 + q] * 36);
          const int e0 = q == pb ? eb : 0;
#pragma unroll
          for (int part = 0; part < 3; ++part) {
            TV val[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) val[k] = src[NV * part + k];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
              const int e = 12 * part + k;           // (compile-time: e / 6, e % 6 are constants)
              const int r = r0 + e / 6, c = c0 + e % 6;
              if (e >= e0 && r >= 0 && r < TILE && c >= 0 && c < TILE) f(r, c, val[k / VEC][k % VEC]);
            }
          }
        }
      } else {
        for (int idx = tid + NT * NPRE; idx < cnt; idx += NT) {
          const int pc = p0 + idx / bb, e = idx % bb;
          const int w = hb.piece_rc[pc];
          const int r = (int)(short)(w >> 16) + e / bd, c = (int)(short)(w & 0xffff) + e % bd;
          if (r >= 0 && r < TILE && c >= 0 && c < TILE) f(r, c, base[(int64_t)hb.piece_blk[pc] * bb + e]);
        }
      }
    }
  }
};
constexpr int HB_NPRE_OFF = 3;    // off-diagonal tiles of a pose graph: <= ~20 pieces (720 elements)

// The pieces of the ADJACENT lower tiles (ti, tj) and (ti, tj + 1) (chol_offdiag2: one workgroup produces both): their runs of
// the piece list are consecutive (tile index ti (ti + 1) / 2 + tj), so they are fetched as ONE run -- both tables of all
// elements in one batch, one exposed round trip, the values in flight until the first gather.  rc: (tile << 16) | (r << 8) | c.
template <typename T, int NPRE>
struct HBPre2 {
  T v[NPRE];
  int rc[NPRE];
  int p0, p1, cnt;   // first piece of tile 0 / of tile 1, elements of both
  int wmeta;         // (HBPre: lane l holds piece_rc of piece l of the run)
  __device__ __forceinline__ void empty() {   // (timing experiments)
    p0 = p1 = cnt = wmeta = 0;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      rc[k] = -1;
      v[k] = T(0);
    }
  }
  __device__ __forceinline__ void to_list(T* list, int tid) const {
#pragma unroll
    for (int k = 0; k < NPRE; ++k)
      if (tid + 256 * k < cnt) list[tid + 256 * k] = v[k];   // (k < NPRE: what the registers hold)
  }
  int tw[NPRE], tblk[NPRE], tpc[NPRE];   // (HBPre: the two phases)
  __device__ __forceinline__ void load_tables(const HBlk& hb, int ti, int tj, int tid) {
    const int bd = hb.bd, bb = bd * bd, t = ti * (ti + 1) / 2 + tj;
    p0 = hb.tile_ptr[t];
    p1 = hb.tile_ptr[t + 1];
    cnt = (hb.tile_ptr[t + 2] - p0) * bb;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      rc[k] = -1;
      v[k] = T(0);
      tw[k] = tblk[k] = tpc[k] = 0;
    }
    wmeta = 0;
    if (cnt <= 0) return;   // (workgroup uniform)
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int idx = tid + 256 * k;
      tpc[k] = p0 + (idx < cnt ? idx : 0) / bb;
      tw[k] = hb.piece_rc[tpc[k]];
      tblk[k] = hb.piece_blk[tpc[k]];
    }
    wmeta = hb.piece_rc[p0 + min(tid & 63, cnt / bb - 1)];
  }
  __device__ __forceinline__ void load_values(const HBlk& hb, int b, int tid) {
    if (cnt <= 0) return;
    const T* base = static_cast<const T*>(hb.blocks) + (int64_t)b * hb.bstride;
    const int bd = hb.bd, bb = bd * bd;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int idx = tid + 256 * k;
      const bool ok = idx < cnt;
      const int e = (ok ? idx : 0) % bb;
      const int r = (int)(short)(tw[k] >> 16) + e / bd, c = (int)(short)(tw[k] & 0xffff) + e % bd;
      const T val = base[(int64_t)tblk[k] * bb + e];
      if (ok && r >= 0 && r < TILE && c >= 0 && c < TILE) {
        rc[k] = ((tpc[k] >= p1 ? 1 : 0) << 16) | (r << 8) | c;
        v[k] = val;
      }
    }
  }
  __device__ __forceinline__ void load(const HBlk& hb, int b, int ti, int tj, int tid) {
    load_tables(hb, ti, tj, tid);
    load_values(hb, b, tid);
  }
  // f(r, c, value) for the pieces of tile ``sel`` (0 / 1)
  template <typename F>
  __device__ __forceinline__ void foreach(const HBlk& hb, int b, int tid, int sel, F&& f) const {
#pragma unroll
    for (int k = 0; k < NPRE; ++k)
      if (rc[k] >= 0 && (rc[k] >> 16) == sel) f((rc[k] >> 8) & 255, rc[k] & 255, v[k]);
    if (cnt > 256 * NPRE) {   // (rare: more pieces than the registers hold)
      const T* base = static_cast<const T*>(hb.blocks) + (int64_t)b * hb.bstride;
      const int bd = hb.bd, bb = bd * bd;
      for (int idx = tid + 256 * NPRE; idx < cnt; idx += 256) {
        const int pc = p0 + idx / bb, e = idx % bb;
        if ((pc >= p1 ? 1 : 0) != sel) continue;
        const int w = hb.piece_rc[pc];
        const int r = (int)(short)(w >> 16) + e / bd, c = (int)(short)(w & 0xffff) + e % bd;
        if (r >= 0 && r < TILE && c >= 0 && c < TILE) f(r, c, base[(int64_t)hb.piece_blk[pc] * bb + e]);
      }
    }
  }
};
constexpr int HB_NPRE_DIAG = 7;   // diagonal tiles: ~21 diagonal blocks + their chain / loop-closure neighbours (~49 pieces)

// ---- H_ij's pieces ADDED to the accumulators by the matrix cores (round 6) ----
// The gather rounds above (zero half a tile of LDS, scatter, barrier, read it back in the accumulator layout, barrier -- 2 rounds
// in fp32, 4 in fp64, 7 / 13 barriers) cost 17 k (fp32) / 40 k (fp64) cycles per tile while the partner workgroup is in its
// K-loop: 2.4 of 44 ms and 5.7 of 93 ms of the headline factorisations (profiles/r6: the same launches with a register-only fake
// of the gather).  A pose graph's off-diagonal tile holds <= ~20 blocks of 6 x 6: 720 values for 16384 accumulators.  So:
// P = -P in registers, the tile's values as ONE contiguous list in LDS (one barrier), and per piece a rank-bd update on the matrix
// cores, acc(tile column c, tile row r) += sum_k [c == c0 + k] V[r - r0][k]: operand A is a 0/1 selector computed from the lane
// index, operand B the piece's values of this lane's row, the accumulator block is picked by wave-uniform branches (the piece's
// origin comes from lane p of ``wmeta`` by readlane).  One exact product v * 1 per element, every other term 0 * x = 0: the result
// is the bit pattern of  v - sum  as before (up to the sign of a zero).
// fp32, v_mfma_f32_32x32x2: lane (rr = lane & 31, g = lane >> 5) supplies A[i = rr][k = g], B[k = g][j = rr]; acc.v[cb][v] is
// D[i = 8 (v / 4) + 4 g + v % 4][j = rr] = tile (row 32 wave + rr, column 32 cb + i)
__device__ __forceinline__ void hb_scatter(Engine<float>::Acc& P, const float* list, int wmeta, int pa, int pb, int bd, int wave,
                                           int lane) {
  const int rr = lane & 31, g = lane >> 5, bb = bd * bd, rw0 = 32 * wave;
  // lane l looks at piece l: does it touch this wave's rows / block cb's columns?  One ballot per accumulator block, then a loop
  // over the set bits -- every loop updates ONE accumulator block (one loop over the pieces with a branch per block made hipcc
  // shuffle the accumulators between registers and spill)
  const int r0l = (int)(short)(wmeta >> 16), c0l = (int)(short)(wmeta & 0xffff);
  const bool rowhit = lane >= pa && lane < pb && r0l + bd > rw0 && r0l < rw0 + 32;
  static_for<4>([&](auto icb) __attribute__((always_inline)) {
    constexpr int cb = decltype(icb)::value;
    unsigned long long mask = __builtin_amdgcn_ballot_w64(rowhit && c0l + bd > 32 * cb && c0l < 32 * cb + 32);
    while (mask) {
      const int p = __builtin_ctzll(mask);
      mask &= mask - 1;
      const int w = __builtin_amdgcn_readlane(wmeta, p);
      const int r0 = (int)(short)(w >> 16), c0 = (int)(short)(w & 0xffff);
      const int dr = rw0 + rr - r0;
      const bool rin = dr >= 0 && dr < bd;
      const float* src = list + p * bb + (rin ? dr : 0) * bd;
      const int t = c0 + g - rr - 32 * cb;   // A[i = rr][k = g], step m: 32 cb + rr == c0 + 2 m + g
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int kc = 2 * m + g;
        const float x = src[min(kc, bd - 1)];
        if (2 * m < bd)
          P.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(t + 2 * m == 0 ? 1.f : 0.f, rin && kc < bd ? x : 0.f, P.v[cb], 0, 0, 0);
      }
    }
  });
}
// fp64, v_mfma_f64_16x16x4: lane (rl = lane & 15, kq = lane >> 4) supplies A[i = rl][k = kq], B[k = kq][j = rl]; acc.v[h][cb][v] is
// D[i = 4 v + kq][j = rl] = tile (row 32 wave + 16 h + rl, column 16 cb + i)
__device__ __forceinline__ void hb_scatter(Engine<double>::Acc& P, const double* list, int wmeta, int pa, int pb, int bd, int wave,
                                           int lane) {
  const int rl = lane & 15, kq = lane >> 4, bb = bd * bd, rw0 = 32 * wave;
  const int r0l = (int)(short)(wmeta >> 16), c0l = (int)(short)(wmeta & 0xffff);
  const bool mine = lane >= pa && lane < pb;
  static_for<2>([&](auto ih) __attribute__((always_inline)) {
    constexpr int h = decltype(ih)::value;
    const int rh0 = rw0 + 16 * h;
    const bool rowhit = mine && r0l + bd > rh0 && r0l < rh0 + 16;
    static_for<8>([&](auto icb) __attribute__((always_inline)) {
      constexpr int cb = decltype(icb)::value;
      unsigned long long mask = __builtin_amdgcn_ballot_w64(rowhit && c0l + bd > 16 * cb && c0l < 16 * cb + 16);
      while (mask) {
        const int p = __builtin_ctzll(mask);
        mask &= mask - 1;
        const int w = __builtin_amdgcn_readlane(wmeta, p);
        const int r0 = (int)(short)(w >> 16), c0 = (int)(short)(w & 0xffff);
        const int dr = rh0 + rl - r0;
        const bool rin = dr >= 0 && dr < bd;
        const double* src = list + p * bb + (rin ? dr : 0) * bd;
        const int t = c0 + kq - rl - 16 * cb;   // A[i = rl][k = kq], step m: 16 cb + rl == c0 + 4 m + kq
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int kc = 4 * m + kq;
          const double x = src[min(kc, bd - 1)];
          if (4 * m < bd)
            P.v[h][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(t + 4 * m == 0 ? 1.0 : 0.0, rin && kc < bd ? x : 0.0, P.v[h][cb], 0, 0, 0);
        }
      }
    });
  });
}

// (the 8-wave fp64 off-diagonal kernel: a wave owns 16 rows of the tile -- acc.v[cb][v] = tile (row 16 wave + rl, column 16 cb + 4 v + kq))
struct Acc16 {
  f64x4 v[8];
};
__device__ __forceinline__ void hb_scatter(Acc16& P, const double* list, int wmeta, int pa, int pb, int bd, int wave, int lane) {
  const int rl = lane & 15, kq = lane >> 4, bb = bd * bd, rh0 = 16 * wave;
  const int r0l = (int)(short)(wmeta >> 16), c0l = (int)(short)(wmeta & 0xffff);
  const bool rowhit = lane >= pa && lane < pb && r0l + bd > rh0 && r0l < rh0 + 16;
  static_for<8>([&](auto icb) __attribute__((always_inline)) {
    constexpr int cb = decltype(icb)::value;
    unsigned long long mask = __builtin_amdgcn_ballot_w64(rowhit && c0l + bd > 16 * cb && c0l < 16 * cb + 16);
    while (mask) {
      const int p = __builtin_ctzll(mask);
      mask &= mask - 1;
      const int w = __builtin_amdgcn_readlane(wmeta, p);
      const int r0 = (int)(short)(w >> 16), c0 = (int)(short)(w & 0xffff);
      const int dr = rh0 + rl - r0;
      const bool rin = dr >= 0 && dr < bd;
      const double* src = list + p * bb + (rin ? dr : 0) * bd;
      const int t = c0 + kq - rl - 16 * cb;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int kc = 4 * m + kq;
        const double x = src[min(kc, bd - 1)];
        if (4 * m < bd)
          P.v[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(t + 4 * m == 0 ? 1.0 : 0.0, rin && kc < bd ? x : 0.0, P.v[cb], 0, 0, 0);
      }
    }
  });
}

// acc += the pieces [lo, hi) of the run ``pre`` describes (HBPre: the tile's, HBPre2: both tiles').  Chunk 0 -- the pieces that sit in
// the registers whole, at most 64 (wmeta) -- goes through ``list`` (written here when ``write_list``: once per run, the caller
// guarantees the buffer is free); a tile with more pieces (rare in a pose graph: > 21 blocks of 6 x 6 in one 128 x 128 tile) takes
// further chunks of 64 straight from memory into ``list + LIST0`` -- two dependent loads and two barriers each, exposed.
// Workgroup uniform control flow; LDS use: LIST0 + 64 bd^2 elements.
constexpr int HB_MODE_SCATTER = 1, HB_MODE_ROUNDS = 2;   // the kernels' HB template argument (0: dense H)
template <typename T, typename Acc, typename Pre, int LIST0, int NT = 256>
__device__ __forceinline__ void hb_add(Acc& P, const Pre& pre, const HBlk& hb, int b, T* list, int lo, int hi, bool write_list,
                                       int tid) {
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int bd = hb.bd, bb = bd * bd;
  const int nreg = min(min(pre.cnt / bb, LIST0 / bb), 64);
  if (write_list) {
    pre.to_list(list, tid);
#ifndef THX_EXP_NO_HBBARRIER   // (timing experiment, WRONG results)
    __syncthreads();
#endif
  }
  hb_scatter(P, list, pre.wmeta, lo, min(hi, nreg), bd, wave, lane);
  if (hi > nreg) {   // (workgroup uniform)
    const T* base = static_cast<const T*>(hb.blocks) + (int64_t)b * hb.bstride;
    T* over = list + LIST0;
    for (int q0 = max(lo, nreg); q0 < hi; q0 += 64) {
      const int nq = min(64, hi - q0);
      __syncthreads();   // the previous chunk has been read
      for (int idx = tid; idx < nq * bb; idx += NT) over[idx] = base[(int64_t)hb.piece_blk[pre.p0 + q0 + idx / bb] * bb + idx % bb];
      const int wm = hb.piece_rc[pre.p0 + q0 + min(lane, nq - 1)];
      __syncthreads();
      hb_scatter(P, over, wm, 0, nq, bd, wave, lane);
    }
  }
}

template <typename T>
struct DiagSmem {
  // ten 32 x LDB sub-blocks; the K-loop's staging buffer (128 x SYRK_LDT) lives in its head
  static constexpr size_t tile = (size_t)10 * 32 * CT<T>::LDB * sizeof(T);
  static_assert(10 * 32 * CT<T>::LDB >= Engine<T>::SYRK_STAGE, "staging buffer must fit in the tile");
  // tile | vvec [128] T | ubuf [32] T | ybuf [ypad] T
  static size_t bytes(int ypad) { return tile + 160 * sizeof(T) + (size_t)ypad * sizeof(T); }
};

template <typename T, bool HB>
__global__ void __launch_bounds__(256, sizeof(T) == 4 ? 3 : 1)
chol_diag_kernel(const T* __restrict__ H, T* __restrict__ L, T* __restrict__ panel, const T* __restrict__ damping,
                 int ellipsoidal, T damping_eps, int32_t* __restrict__ info, int n, int64_t ld, int j0, int ntiles,
                 const T* __restrict__ rhs, T* __restrict__ yout, int64_t ldv, TilePat pat, HBlk hb) {
  using C = CT<T>;
  using V = typename C::V;
  using E = Engine<T>;
  const int j = j0 + blockIdx.y;   // (level schedule: blockIdx.y runs over the level's block columns)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);  // lower sub-blocks (tblk); its head doubles as the K-loop staging buffer
  T* vvec = reinterpret_cast<T*>(smem_raw + DiagSmem<T>::tile);
  T* ubuf = vvec + 128;
  T* ybuf = ubuf + 32;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const LFrame lf = lframe(pat, ld);
  const int64_t mat = (int64_t)b * ld * ld;            // H: always the dense frame (or the block list)
  const int64_t lmat = (int64_t)b * lf.pstride;        // L: dense frame or tile-packed
  const int64_t ldt = lf.ld;
  T* const Ljj = L + lmat + lf.tile(j, j, j);          // the diagonal tile of L
  const int row0 = j * TILE;
  const int valid = tile_rows(pat, n, j);

  const bool fwd = rhs != nullptr;
#ifdef THX_DIAG_PROF
  long long stamps[24];
  int nst = 0;
#define THX_STAMP() stamps[nst++] = (long long)__builtin_readcyclecounter()
#else
#define THX_STAMP()
#endif
  THX_STAMP();

  // SYRK on the 36 lower 16x16 blocks of the tile, nine per wave (Engine<T>::syrk36)
  // tile-sparse: only the block columns k < j in which row panel j is non-zero
  const int32_t* klist = pat.diag_k ? pat.diag_k + pat.diag_kptr[j] : nullptr;
  const int Kspan = pat.rl ? (pat.rl_la ? TILE : 0) : (pat.diag_k ? (pat.diag_kptr[j + 1] - pat.diag_kptr[j]) * TILE : row0);
  const int kcol0 = (pat.rl && pat.rl_la) ? (j - 1) * TILE : 0;   // (right-looking look-ahead: the K-loop is the one tile L_j,j-1)
  const bool ycompact = pat.ent_col != nullptr;   // (level schedule: ybuf holds the K-list's blocks of y only)
  typename E::Sy acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[i][k] = T(0);
  T tpart = T(0);  // this thread's half of (L_j,0:j y)[tid >> 1]
  std::conditional_t<sizeof(T) == 4, float4, f64x4> hpre[9];
  // issued behind the loads of the first k-chunk (kloop_f's after_issue hook): y_0:j of the earlier columns -> LDS, and the
  // H_jj blocks, in flight during the whole K-loop
  HBPre<T, HB ? HB_NPRE_DIAG : 1> hbp;
  auto prologue = [&]() __attribute__((always_inline)) {
    if (fwd) {
      if (ycompact) {   // level schedule: the blocks of y this column's K-list names, back to back
        for (int k = tid; k < Kspan; k += 256) ybuf[k] = yout[(int64_t)b * ldv + klist[k >> 7] * TILE + (k & (TILE - 1))];
      } else {
        for (int k = tid; k < row0; k += 256) ybuf[k] = yout[(int64_t)b * ldv + k];
      }
    }
    if constexpr (!HB) {
      const T* Hjj = H + mat + (int64_t)row0 * ld + row0;
      if (wave == 0) E::template syrk36_prefetch<0>(Hjj, ld, valid, hpre, lane);
      else if (wave == 1) E::template syrk36_prefetch<1>(Hjj, ld, valid, hpre, lane);
      else if (wave == 2) E::template syrk36_prefetch<2>(Hjj, ld, valid, hpre, lane);
      else E::template syrk36_prefetch<3>(Hjj, ld, valid, hpre, lane);
    } else {   // block-compact H: the tile's blocks are ADDED after the SYRK (below); the "H" of the store is zero
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) reinterpret_cast<T*>(&hpre[i])[k] = T(0);
      hbp.load(hb, b, j, j, tid);
    }
  };
#ifdef THX_OFF_PROLOGUE_FIRST   // the round-1 order (A/B timing)
  prologue();
#endif
  kloop_f<T, true, true, E::SYRK_LDT, (sizeof(T) == 8 && CT<T>::KB == 32)>(
      L + lmat + (lf.packed ? 0 : (int64_t)row0 * ld) + kcol0, valid, nullptr, 0, ldt, Kspan, tile, nullptr, tid,
      (fwd && !pat.rl) ? ybuf : nullptr, &tpart, [&]() __attribute__((always_inline)) {
    if (wave == 0) E::template syrk36<0>(tile, acc, lane);
    else if (wave == 1) E::template syrk36<1>(tile, acc, lane);
    else if (wave == 2) E::template syrk36<2>(tile, acc, lane);
    else E::template syrk36<3>(tile, acc, lane);
  },
#ifdef THX_OFF_PROLOGUE_FIRST
  NoHook{}, klist, lf.packed ? pat.diag_s + pat.diag_kptr[j] : nullptr, nullptr, lf.pstride, ycompact);
#else
  prologue, klist, lf.packed ? pat.diag_s + pat.diag_kptr[j] : nullptr, nullptr, lf.pstride, ycompact);
#endif

  // ---- S = H_jj (+ damping on the diagonal) - acc -> LDS tile; identity padding outside the matrix ----
  __syncthreads();  // staging buffer is free
  THX_STAMP();
  if (tid < TILE) vvec[tid] = (fwd && tid < valid) ? rhs[(int64_t)b * ldv + row0 + tid] : T(0);
  {
    const bool damp = damping != nullptr;
    const T lam = damp ? damping[b] : T(0);
    const bool sd = damp && !HB;   // (block-compact H: the damping rides on the diagonal elements of the gathered blocks)
    if (wave == 0) E::template syrk36_store<0>(tile, hpre, acc, lane, valid, sd, lam, ellipsoidal, damping_eps);
    else if (wave == 1) E::template syrk36_store<1>(tile, hpre, acc, lane, valid, sd, lam, ellipsoidal, damping_eps);
    else if (wave == 2) E::template syrk36_store<2>(tile, hpre, acc, lane, valid, sd, lam, ellipsoidal, damping_eps);
    else E::template syrk36_store<3>(tile, hpre, acc, lane, valid, sd, lam, ellipsoidal, damping_eps);
    __syncthreads();  // vvec visible
    {  // g_j - L_j,0:j y : thread pair (2r, 2r+1) holds the two halves of row r's sum
      const T tsum = tpart + __shfl_xor(tpart, 1);
      if (fwd && (tid & 1) == 0) vvec[tid >> 1] -= tsum;
    }
    if constexpr (HB) {   // S += H_jj (+ damping): each element of the tile's lower triangle belongs to at most one piece
      hbp.foreach(hb, b, tid, [&](int r, int c, T v) __attribute__((always_inline)) {
        if (c > r) return;
        if (r == c && damp) v = ellipsoidal ? v + (lam * v + damping_eps) : v + lam;
        tile[tblk<T>(r >> 5, c >> 5) + (r & 31) * C::LDB + (c & 31)] += v;
      });
    }
  }
  __syncthreads();
  THX_STAMP();

  // ---- blocked right-looking Cholesky on the LDS tile, 32-wide sub-blocks.  Afterwards the tile IS the
  //      solve panel: W_ss = L_ss^-1 on the diagonal sub-blocks, -L_us below them. ----
#if defined(THX_SERIAL_WAVE_BID)
  const int sw = (blockIdx.x >> 8) & 3;
#else
  const int sw = 0;
#endif
  for (int sb = 0; sb < 4; ++sb) {
    T* Dss = tile + tblk<T>(sb, sb);
#ifndef THX_POTRF_FLAT
    if (wave == sw) {
      THX_STAMP();
      const int bad = potrf_inv32<T>(Dss, Ljj + (int64_t)(32 * sb) * ldt + 32 * sb, ldt, valid - 32 * sb, lane);
      if (bad != 0 && lane == 0 && info[b] == 0) info[b] = row0 + 32 * sb + bad;
      THX_STAMP();
    }
#else  // the unblocked 32-step chain (kept for A/B timing)
    if (wave == sw) {
      T a[32];
      {
        const V* rp = reinterpret_cast<const V*>(Dss + (lane & 31) * C::LDB);
#pragma unroll
        for (int q = 0; q < 32 / C::VEC; ++q) {
          const V v = rp[q];
          if constexpr (sizeof(T) == 4) {
            a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
          } else {
            a[2 * q] = v.x; a[2 * q + 1] = v.y;
          }
        }
      }
      THX_STAMP();
      const int bad = potrf32<T>(a);
      THX_STAMP();
      if (bad != 0 && lane == 0 && info[b] == 0) info[b] = row0 + 32 * sb + bad;
      // L_ss straight to global memory: one 32-element row per lane, zeros above the diagonal
      const int lr = lane & 31, grow = 32 * sb + lr;
      if (lane < 32 && grow < valid) {
        V* gp = reinterpret_cast<V*>(Ljj + (int64_t)grow * ldt + 32 * sb);
#pragma unroll
        for (int q = 0; q < 32 / C::VEC; ++q) {
          if constexpr (sizeof(T) == 4) {
            gp[q] = make_float4(4 * q <= lr ? a[4 * q] : 0.f, 4 * q + 1 <= lr ? a[4 * q + 1] : 0.f,
                                4 * q + 2 <= lr ? a[4 * q + 2] : 0.f, 4 * q + 3 <= lr ? a[4 * q + 3] : 0.f);
          } else {
            gp[q] = make_double2(2 * q <= lr ? a[2 * q] : 0.0, 2 * q + 1 <= lr ? a[2 * q + 1] : 0.0);
          }
        }
      }
      // L_ss back into its LDS block (zeros above the diagonal), then invert it from there
      {
        V* rp = reinterpret_cast<V*>(Dss + lr * C::LDB);
        if (lane < 32) {
#pragma unroll
          for (int q = 0; q < 32 / C::VEC; ++q) {
            if constexpr (sizeof(T) == 4) {
              rp[q] = make_float4(4 * q <= lr ? a[4 * q] : 0.f, 4 * q + 1 <= lr ? a[4 * q + 1] : 0.f,
                                  4 * q + 2 <= lr ? a[4 * q + 2] : 0.f, 4 * q + 3 <= lr ? a[4 * q + 3] : 0.f);
            } else {
              rp[q] = make_double2(2 * q <= lr ? a[2 * q] : 0.0, 2 * q + 1 <= lr ? a[2 * q + 1] : 0.0);
            }
          }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's LDS writes are visible to its reads
        __builtin_amdgcn_wave_barrier();
      }
      using WA = std::conditional_t<sizeof(T) == 8, double, float>;
      WA w[32];
      THX_STAMP();
      inv32<T, WA, C::LDB>(Dss, w, lane);
      THX_STAMP();
      __builtin_amdgcn_wave_barrier();
      if (lane < 32) {
#pragma unroll
        for (int i = 0; i < 32; ++i) Dss[i * C::LDB + lr] = (T)w[i];  // W[i][lr]; zero for i < lr
      }
    }
#endif
    __syncthreads();
    if (sb == 3) break;
    // L_us = S_us W_ss^T for the sub-blocks below (one per wave), stored negated
    {
      const int u = sb + 1 + wave;
      if (u < 4) {
        T* Dus = tile + tblk<T>(u, sb);
        typename E::Blk X;
        E::blk_zero(X);
        E::blk_mma(Dss, Dus, X, lane, T(1));
        E::blk_store(X, Dus, lane, T(-1));
      }
    }
    __syncthreads();
    // trailing update S_uv -= L_us L_vs^T, sb < v <= u: blocks dealt round-robin to the waves
    {
      int idx = 0;
      for (int u = sb + 1; u < 4; ++u)
        for (int v = sb + 1; v <= u; ++v, ++idx) {
          if ((idx & 3) != wave) continue;
          T* Duv = tile + tblk<T>(u, v);
          typename E::Blk D;
          E::blk_load(D, Duv, lane);
          // tile(v,s) = -L_vs is negated on load, tile(u,s) = -L_us:  D += (+L_vs)(-L_us)^T
          E::blk_mma(tile + tblk<T>(v, sb), tile + tblk<T>(u, sb), D, lane, T(-1));
          E::blk_store(D, Duv, lane, T(1));
        }
    }
    __syncthreads();
  }

  THX_STAMP();
  // ---- outputs: strictly-lower sub-blocks of L_jj (= -tile), the panel, y_j ----
  {  // (the panel's sub-blocks above the diagonal are never read -- chol_offdiag and the solves use the lower ten -- and
     //  are not written)
    constexpr int VPR = 32 / C::VEC;  // vectors per sub-block row
    T* Lt = Ljj;
    T* P = panel + ((int64_t)b * ntiles + j) * TILE * TILE;
    for (int u = 0; u < 4; ++u)
      for (int v = 0; v <= u; ++v) {
        const T* blk = tile + tblk<T>(u, v);
#pragma unroll
        for (int idx = tid; idx < 32 * VPR; idx += 256) {
          const int rr = idx / VPR, c = (idx % VPR) * C::VEC;
          const V val = *reinterpret_cast<const V*>(blk + rr * C::LDB + c);
          *reinterpret_cast<V*>(P + (32 * u + rr) * TILE + 32 * v + c) = val;
          if (u > v && 32 * u + rr < valid) {
            V* dst = reinterpret_cast<V*>(Lt + (int64_t)(32 * u + rr) * ldt + 32 * v + c);
            if constexpr (sizeof(T) == 4) *dst = make_float4(-val.x, -val.y, -val.z, -val.w);
            else *dst = make_double2(-val.x, -val.y);
          }
        }
      }
  }
  if (fwd) {
    if (wave == 0) {   // (the shared column-by-column substitution: same rounding as chol_potrf_kernel's)
      for (int sb = 0; sb < 4; ++sb) {
        typename E::Blk Wb;
        E::blk_load(Wb, tile + tblk<T>(sb, sb), lane);
        fwd_diag_block<T>(Wb, vvec, sb, lane);
        for (int u = sb + 1; u < 4; ++u) {
          typename E::Blk Xb;
          E::blk_load(Xb, tile + tblk<T>(u, sb), lane);
          fwd_below_block<T>(Xb, vvec, sb, u, lane);
        }
        wave_lds_fence();
      }
    }
    __syncthreads();
    if (tid < valid) yout[(int64_t)b * ldv + row0 + tid] = vvec[tid];
  }
#ifdef THX_DIAG_PROF
  THX_STAMP();
  __syncthreads();
  if (tid == sw * 64) {
    T* P = panel + ((int64_t)b * ntiles + j) * TILE * TILE;
    for (int k = 1; k < nst; ++k) P[k] = (T)(stamps[k] - stamps[0]);
    P[0] = (T)nst;
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// The diagonal phase SPLIT in two kernels (default; THX_CHOL_FUSED_DIAG=1 selects chol_diag_kernel above):
//   chol_syrk_kernel : the MFMA half of chol_diag -- S = H_jj + damping - L_j,0:j L_j,0:j^T (36 lower 16x16 blocks, nine per wave),
//                      riding on it g_j - L_j,0:j y -- written to the diagonal tile's place in the global factor / to y_j.
//                      No serial phase: all four waves of all three resident workgroups issue MFMAs for the kernel's whole life.
//   chol_potrf_kernel: the serial half, ONE WAVE per tile.  The tile's ten lower 32x32 sub-blocks live in that wave's registers in
//                      the MFMA C/D layout (160 VGPRs in fp32); the sub-block TRSMs and trailing updates are register x register
//                      MFMAs (Engine::blk_mma_rr: no LDS traffic at all), only the 32x32 diagonal sub-block being factorised and
//                      inverted passes through a 4.6 KB LDS block (potrf_inv32).  5 KB of LDS and <= 256 VGPRs per tile:
//                      EIGHT tiles per CU are in their latency-bound pivot chains at once, against three with chol_diag -- whose
//                      workgroup pinned 53 KB of LDS and three idle waves' registers for the 126 k cycles of its chain, i.e. kept
//                      a third of a CU from anything else.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct SyrkSmem {
  static size_t bytes(int ypad) { return (size_t)Engine<T>::SYRK_STAGE * sizeof(T) + (size_t)ypad * sizeof(T); }
};

#ifndef THX_SYRK32_WAVES
#define THX_SYRK32_WAVES 3
#endif
#ifndef THX_SYRK64_WAVES
#define THX_SYRK64_WAVES 3   // waves per SIMD the fp64 block-compact SYRK is compiled for: 3 (168 VGPRs + 20 B of scratch) instead of 2
                             // (200 VGPRs) takes 0.7 ms off the headline factorisation (87.9 / 88.0 -> 87.2 / 87.3 ms: its short
                             // K-loops want a third workgroup per CU); 4 (128 VGPRs, 168 B of scratch) costs 2.7 ms (profiles/r6/ag_)
#endif
template <typename T, bool HB>
__global__ void __launch_bounds__(256, sizeof(T) == 4 ? (HB ? THX_SYRK32_WAVES : 3) : (HB ? THX_SYRK64_WAVES : 2))
chol_syrk_kernel(const T* __restrict__ H, T* __restrict__ L, const T* __restrict__ damping, int ellipsoidal, T damping_eps,
                 int n, int64_t ld, int j0, const T* __restrict__ rhs, T* __restrict__ yout, int64_t ldv, TilePat pat, HBlk hb) {
  using E = Engine<T>;
  const int j = j0 + blockIdx.y;   // (level schedule: blockIdx.y runs over the level's block columns)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* stage = reinterpret_cast<T*>(smem_raw);                 // K-loop staging buffer, 128 x SYRK_LDT
  T* ybuf = stage + E::SYRK_STAGE;                           // y_0:j of the earlier columns
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const LFrame lf = lframe(pat, ld);
  const int64_t mat = (int64_t)b * ld * ld;            // H (dense frame)
  const int64_t lmat = (int64_t)b * lf.pstride;        // L (dense frame or tile-packed)
  const int64_t ldt = lf.ld;
  const int row0 = j * TILE;
  const int valid = tile_rows(pat, n, j);
  const bool fwd = rhs != nullptr;

  // tile-sparse: only the block columns k < j in which row panel j is non-zero
  const int32_t* klist = pat.diag_k ? pat.diag_k + pat.diag_kptr[j] : nullptr;
  const int Kspan = pat.diag_k ? (pat.diag_kptr[j + 1] - pat.diag_kptr[j]) * TILE : row0;
  const bool ycompact = pat.ent_col != nullptr;   // (level schedule: ybuf holds the K-list's blocks of y only)
  typename E::Sy acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[i][k] = T(0);
  T tpart = T(0);  // this thread's half of (L_j,0:j y)[tid >> 1]
  std::conditional_t<sizeof(T) == 4, float4, f64x4> hpre[9];
  HBPre<T, HB ? HB_NPRE_DIAG : 1> hbp;
  auto prologue = [&]() __attribute__((always_inline)) {
    if (fwd) {
      if (ycompact) {   // level schedule: the blocks of y this column's K-list names, back to back
        for (int k = tid; k < Kspan; k += 256) ybuf[k] = yout[(int64_t)b * ldv + klist[k >> 7] * TILE + (k & (TILE - 1))];
      } else {
        for (int k = tid; k < row0; k += 256) ybuf[k] = yout[(int64_t)b * ldv + k];
      }
    }
    if constexpr (!HB) {
      const T* Hjj = H + mat + (int64_t)row0 * ld + row0;
      if (wave == 0) E::template syrk36_prefetch<0>(Hjj, ld, valid, hpre, lane);
      else if (wave == 1) E::template syrk36_prefetch<1>(Hjj, ld, valid, hpre, lane);
      else if (wave == 2) E::template syrk36_prefetch<2>(Hjj, ld, valid, hpre, lane);
      else E::template syrk36_prefetch<3>(Hjj, ld, valid, hpre, lane);
    } else {   // block-compact H: the tile's blocks are ADDED after the SYRK (below); the "H" of the store is zero
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) reinterpret_cast<T*>(&hpre[i])[k] = T(0);
      hbp.load(hb, b, j, j, tid);
    }
  };
  kloop_f<T, true, true, E::SYRK_LDT, (sizeof(T) == 8 && CT<T>::KB == 32)>(
      L + lmat + (lf.packed ? 0 : (int64_t)row0 * ld), valid, nullptr, 0, ldt, Kspan, stage, nullptr, tid,
      fwd ? ybuf : nullptr, &tpart, [&]() __attribute__((always_inline)) {
    if (wave == 0) E::template syrk36<0>(stage, acc, lane);
    else if (wave == 1) E::template syrk36<1>(stage, acc, lane);
    else if (wave == 2) E::template syrk36<2>(stage, acc, lane);
    else E::template syrk36<3>(stage, acc, lane);
  }, prologue, klist, lf.packed ? pat.diag_s + pat.diag_kptr[j] : nullptr, nullptr, lf.pstride, ycompact);

  {
    const bool damp = damping != nullptr;
    const T lam = damp ? damping[b] : T(0);
    T* Lt = L + lmat + lf.tile(j, j, j);
    const bool sd = damp && !HB;
    if (wave == 0) E::template syrk36_store_global<0>(Lt, ldt, hpre, acc, lane, valid, sd, lam, ellipsoidal, damping_eps);
    else if (wave == 1) E::template syrk36_store_global<1>(Lt, ldt, hpre, acc, lane, valid, sd, lam, ellipsoidal, damping_eps);
    else if (wave == 2) E::template syrk36_store_global<2>(Lt, ldt, hpre, acc, lane, valid, sd, lam, ellipsoidal, damping_eps);
    else E::template syrk36_store_global<3>(Lt, ldt, hpre, acc, lane, valid, sd, lam, ellipsoidal, damping_eps);
    if constexpr (HB) {
      // block-compact H: the tile now holds -L_j L_j^T; its pieces of H (+ damping on the diagonal) are added in place.  The
      // stores above are this workgroup's own: visible to all its threads after the fence + barrier.
      __threadfence_block();
      __syncthreads();
      hbp.foreach(hb, b, tid, [&](int r, int c, T v) __attribute__((always_inline)) {
        if (c > r || r >= valid) return;
        if (r == c && damp) v = ellipsoidal ? v + (lam * v + damping_eps) : v + lam;
        Lt[(int64_t)r * ldt + c] += v;
      });
    }
  }
  if (fwd) {  // g_j - L_j,0:j y -> y_j's place (chol_potrf_kernel finishes it): thread pair (2r, 2r+1) holds the halves of row r
    const T tsum = tpart + __shfl_xor(tpart, 1);
    const int r = tid >> 1;
    if ((tid & 1) == 0 && r < valid) yout[(int64_t)b * ldv + row0 + r] = rhs[(int64_t)b * ldv + row0 + r] - tsum;
  }
}

__device__ __forceinline__ constexpr int bidx(int u, int v) { return u * (u + 1) / 2 + v; }

#ifndef THX_POTRF_F64_WAVES_PER_SIMD
#define THX_POTRF_F64_WAVES_PER_SIMD 1   // (experiment knob: 2 = at most 256 registers, the compiler spills the rest to scratch)
#endif
template <typename T>
__global__ void __launch_bounds__(64, sizeof(T) == 4 ? 2 : THX_POTRF_F64_WAVES_PER_SIMD)
chol_potrf_kernel(T* __restrict__ L, T* __restrict__ panel, int32_t* __restrict__ info, int n, int64_t pstride, int64_t tile_off0,
                  int64_t ld, int j0, int ntiles, T* __restrict__ yout, int64_t ldv, const int32_t* __restrict__ tile_valid) {
  // (the diagonal tile of problem b starts at L + b * pstride + tile_off, row stride ld: dense frame or tile-packed factor;
  //  level schedule -- tile-packed factor only -- blockIdx.y runs over the level's block columns: slot j0 + blockIdx.y)
  const int j = j0 + blockIdx.y;
  const int64_t tile_off = tile_off0 + (int64_t)blockIdx.y * TILE * TILE;
  using C = CT<T>;
  using E = Engine<T>;
  using Blk = typename E::Blk;
  __shared__ __attribute__((aligned(16))) T Dss[32 * C::LDB];   // the diagonal sub-block being factorised / inverted
  __shared__ __attribute__((aligned(16))) T vvec[TILE];         // right-hand side / solution of the fused forward substitution
  // fp32, two waves per SIMD (256 VGPRs): while the FIRST diagonal sub-block is factorised -- nine other sub-blocks live next to
  // the temporaries of potrf_inv32 -- the last block row waits in LDS instead of in spilled registers
  constexpr int PARK = sizeof(T) == 4 ? 3 : 0;
  __shared__ __attribute__((aligned(16))) T park[PARK > 0 ? PARK * 32 * C::LDB : 4];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int row0 = j * TILE, valid = tile_valid ? tile_valid[j] : min(TILE, n - row0);
  T* Lt = L + (int64_t)b * pstride + tile_off;
  const bool fwd = yout != nullptr;

  // the tile: S (written by chol_syrk_kernel) -> registers; identity outside the matrix
  Blk Tb[10];
  static_for<4>([&](auto iu) __attribute__((always_inline)) {
    constexpr int u = decltype(iu)::value;
    static_for<u + 1>([&](auto iv) __attribute__((always_inline)) {
      constexpr int v = decltype(iv)::value;
      E::blk_load_global(Tb[bidx(u, v)], Lt + (int64_t)(32 * u) * ld + 32 * v, Lt, ld, valid - 32 * u, valid - 32 * v, u == v, lane);
    });
  });
  if (fwd) {
#pragma unroll
    for (int k = lane; k < TILE; k += 64) vvec[k] = k < valid ? yout[(int64_t)b * ldv + row0 + k] : T(0);
  }

  // blocked right-looking Cholesky over the four 32-wide sub-block columns.  A finished column -- W_ss = L_ss^-1 on the diagonal
  // sub-block, -L_us below it: the solve panel's column -- is used at once for the fused forward substitution and stored, so
  // its registers are free for the rest of the factorisation.
  T* P = panel + ((int64_t)b * ntiles + j) * TILE * TILE;
  static_for<4>([&](auto isb) __attribute__((always_inline)) {
    constexpr int sb = decltype(isb)::value;
    E::blk_store(Tb[bidx(sb, sb)], Dss, lane, T(1));
    if constexpr (sb == 0 && PARK > 0) {
      static_for<PARK>([&](auto ik) __attribute__((always_inline)) {
        constexpr int k = decltype(ik)::value;
        E::blk_store(Tb[bidx(3, 1 + k)], park + k * 32 * C::LDB, lane, T(1));
      });
    }
    wave_lds_fence();
    const int bad = potrf_inv32<T>(Dss, Lt + (int64_t)(32 * sb) * ld + 32 * sb, ld, valid - 32 * sb, lane);
    if (bad != 0 && lane == 0 && info[b] == 0) info[b] = row0 + 32 * sb + bad;
    wave_lds_fence();
    E::blk_load(Tb[bidx(sb, sb)], Dss, lane);   // W_ss (full 32 x 32, zero above the diagonal)
    if constexpr (sb == 0 && PARK > 0) {
      static_for<PARK>([&](auto ik) __attribute__((always_inline)) {
        constexpr int k = decltype(ik)::value;
        E::blk_load(Tb[bidx(3, 1 + k)], park + k * 32 * C::LDB, lane);
      });
    }
    E::blk_store_global(Tb[bidx(sb, sb)], P + (32 * sb) * TILE + 32 * sb, TILE, 32, lane, T(1));
    if (fwd) fwd_diag_block<T>(Tb[bidx(sb, sb)], vvec, sb, lane);   // y_s = W_ss u_s (u_s: what the earlier columns left)
    // L_us = S_us W_ss^T for the sub-blocks below, kept negated; u_u += (-L_us) y_s
    static_for<3 - sb>([&](auto iu) __attribute__((always_inline)) {
      constexpr int u = sb + 1 + decltype(iu)::value;
      Blk X;
      E::blk_zero(X);
      E::blk_mma_rr(Tb[bidx(sb, sb)], Tb[bidx(u, sb)], X, T(1));
      E::blk_neg(X);
      Tb[bidx(u, sb)] = X;
      E::blk_store_global(X, P + (32 * u) * TILE + 32 * sb, TILE, 32, lane, T(1));
      E::blk_store_global(X, Lt + (int64_t)(32 * u) * ld + 32 * sb, ld, valid - 32 * u, lane, T(-1));
      if (fwd) fwd_below_block<T>(X, vvec, sb, u, lane);
    });
    // trailing update S_uv -= L_us L_vs^T, sb < v <= u  (Tb(v,sb) = -L_vs is negated back, Tb(u,sb) = -L_us)
    static_for<3 - sb>([&](auto iu) __attribute__((always_inline)) {
      constexpr int u = sb + 1 + decltype(iu)::value;
      static_for<u - sb>([&](auto iv) __attribute__((always_inline)) {
        constexpr int v = sb + 1 + decltype(iv)::value;
        E::blk_mma_rr(Tb[bidx(v, sb)], Tb[bidx(u, sb)], Tb[bidx(u, v)], T(-1));
      });
    });
    if (fwd) wave_lds_fence();   // the vvec updates of this column before the next column reads them
  });
  if (fwd) {
#pragma unroll
    for (int k = lane; k < TILE; k += 64)
      if (k < valid) yout[(int64_t)b * ldv + row0 + k] = vvec[k];
  }
}

// ------------------------------------------------------------------------------------------------
// chol_offdiag: GEMM K-loop + blocked MFMA substitution  L_ij = (H_ij - sum) L_jj^-T.
// Two kernels, one per dtype; both keep two workgroups per CU resident and move H / the result between global memory and
// registers directly.  (The first version of this kernel staged H, the full 128x130 panel and the result through one LDS
// tile: 67 / 133 KB, every load phase exposed; it is gone.)
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// chol_offdiag, fp32.  Nothing of the epilogue waits on memory: the H tile is prefetched into registers in the accumulator layout and the ten
// lower sub-blocks of the panel (40 KB, XOR-swizzled so that unpadded 32x32 blocks read conflict
// free) are copied to LDS BEFORE the K-loop; the result is stored straight from the registers.
// LDS: staging 36 KB + panel 40 KB -> two workgroups per CU.
// ------------------------------------------------------------------------------------------------
constexpr int OFF32_STAGE_FLOATS = 2 * 128 * 36;
constexpr int OFF32_SMEM = (OFF32_STAGE_FLOATS + 10 * 1024) * 4;

// D.block(S) += Pc[block (S,Tt)] * Bs.block(Tt)^T with the swizzled compact panel, i.e. for every tile row r this wave
// owns:  D[r][32S + i] += sum_{c in block Tt} M[32S + i][c] * Bs[r][c].  The B operand is the accumulator itself: an MFMA's
// k index is only a pairing of columns (k = 0/1 <-> columns c and c+4 held by lane groups 0/1).
template <int S, int Tt>
__device__ __forceinline__ void sub_mma_sw(const float* Pc, const Engine<float>::Acc& Bs, Engine<float>::Acc& D,
                                           int lane) {
  const int rl = lane & 31, g = lane >> 5;
  const float* brow = Pc + (S * (S + 1) / 2 + Tt) * 1024 + rl * 32;
  const int sw = (rl >> 1) & 7;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 fa = *reinterpret_cast<const float4*>(brow + (((2 * q + g) ^ sw) << 2));
    D.v[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, Bs.v[Tt][4 * q + 0], D.v[S], 0, 0, 0);
    D.v[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, Bs.v[Tt][4 * q + 1], D.v[S], 0, 0, 0);
    D.v[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, Bs.v[Tt][4 * q + 2], D.v[S], 0, 0, 0);
    D.v[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, Bs.v[Tt][4 * q + 3], D.v[S], 0, 0, 0);
  }
}

template <int HB>   // 0: dense H; HB_MODE_SCATTER / HB_MODE_ROUNDS: block-compact H, how a tile's pieces reach the accumulators
__global__ void __launch_bounds__(256, 2)
chol_offdiag_f32_kernel(const float* __restrict__ H, float* __restrict__ L, const float* __restrict__ panel, int n,
                        int64_t ld, int jarg, int ntiles, int i_first, int nrow_tiles, int B, TilePat pat, HBlk hb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  // Default map: the problem is the slow index -- all row tiles of a problem run at the same time on ONE XCD and share the column
  // panel in its L2.  pat.lpt (small-batch tile-sparse launches, the look-ahead schedule): the ENTRY is the slow index, i.e. the
  // column's entries are dispatched longest K-list first (a band's entries are sorted that way: the nearer the diagonal, the
  // longer) -- with two to three rounds of workgroups per launch the tail of a LONG tile started last costs more than the panel
  // re-reads (same-box A/B profiles/r4/s_: banded BA system 6.62 -> 6.34 / 6.46 ms, bit-identical factor).
  const int b8 = gridDim.x / (8 * nrow_tiles);
  const int b = pat.lpt ? (slot % b8) * 8 + xcd : (slot / nrow_tiles) * 8 + xcd;
  const int rslot = pat.lpt ? slot / b8 : slot % nrow_tiles;
  // row tiles [i_first, i_first + nrow_tiles) of block column j -- or, tile-sparse, entries [i_first, i_first + nrow_tiles) of the
  // column's list of non-zero row tiles
  // (tile-sparse: i_first = first ENTRY of the launch, relative to the column's list -- level schedule: absolute, and the entry
  //  names its block column)
  const int ent = pat.col_row ? (pat.ent_col ? 0 : pat.col_ptr[jarg]) + i_first + rslot : 0;
  // right-looking trailing update of block column jc (pat.rl = 2 + jc; dense frames): slot t -> tile (i, k), jc < k <= i,
  // rows of the lower triangle numbered row by row; "j" is the tile's own block column k, the K-loop is the one tile jc
  const bool combo = pat.rl_nsub > 0;   // (TilePat.rl_nsub: substitution tiles of column jarg + update tiles of column jarg - 1)
  const bool upd = combo ? rslot >= pat.rl_nsub : pat.rl >= 2;
  const int jc = combo ? jarg - 1 : pat.rl - 2;
  const int ub = combo ? jarg + 1 : jc + 1;          // first block row / column of the updated tiles
  const bool sla = pat.rl == 1 && pat.rl_la != 0;    // substitution tile with the previous column's update as a one-tile K-loop
  int ui = 0, uk = 0;
  if (upd) {   // (i_first: the launch's first slot)
    const int us = combo ? rslot - pat.rl_nsub : rslot + i_first;
    ui = (int)((__builtin_sqrtf(8.f * (float)us + 1.f) - 1.f) * 0.5f);
    while ((ui + 1) * (ui + 2) / 2 <= us) ++ui;
    while (ui * (ui + 1) / 2 > us) --ui;
    uk = us - ui * (ui + 1) / 2;
  }
  const int j = upd ? ub + uk : (pat.ent_col ? pat.ent_col[ent] : jarg);
  const int i = upd ? ub + ui : (pat.col_row ? pat.col_row[ent] : i_first + rslot);
  const int32_t* klist = pat.col_row ? pat.tile_k + pat.tile_kptr[ent] : nullptr;
  const int Kspan = pat.rl ? ((upd || sla) ? TILE : 0) : (pat.col_row ? (pat.tile_kptr[ent + 1] - pat.tile_kptr[ent]) * TILE : j * TILE);
  const int kcol0 = upd ? jc * TILE : (sla ? (jarg - 1) * TILE : 0);   // first column of the K-loop inside the row panels
  if (b >= B) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const LFrame lf = lframe(pat, ld);
  const int64_t mat = (int64_t)b * ld * ld;            // H (dense frame)
  const int64_t lmat = (int64_t)b * lf.pstride;        // L (dense frame or tile-packed)
  const int64_t ldt = lf.ld;
  float* const Lij = L + lmat + lf.tile(i, j, ntiles + ent);   // the tile this workgroup produces
  const int32_t* ksa = lf.packed ? pat.tile_sa + pat.tile_kptr[ent] : nullptr;
  const int32_t* ksb = lf.packed ? pat.tile_sb + pat.tile_kptr[ent] : nullptr;
  const int col0 = j * TILE, row0 = i * TILE;
  const int validB = tile_rows(pat, n, i);  // (columns of tile j beyond the matrix -- last tile / per-tile padding -- come out as exact zeros)
  float* sA = smem;
  float* sB = smem + 128 * 36;
  float* Pc = smem + OFF32_STAGE_FLOATS;

#ifdef THX_OFF_PROF  // timing build (tools/prof/off_prof.py): cycle stamps overwrite the head of the output tile
  long long st[7];
  st[0] = (long long)__builtin_readcyclecounter();
  st[5] = 0;
  const long long wc0 = (long long)wall_clock64();
#endif
  // ---- prefetch: panel sub-blocks (s,t), t <= s, then the H tile.  Issued from inside the K-loop's prologue, AFTER the
  //      loads of the first k-chunk: one exposed memory latency per workgroup instead of two (in-kernel stamps: 8.8-11.2 k
  //      cycles from kernel entry to the first MFMA, profiles/r2/a_offdiag_stamps.txt) ----
  // (a "lean" variant without any prefetch -- 40 KB LDS, 168 VGPRs, three workgroups per CU -- measured 1-2 % SLOWER:
  //  the K-loop's 82 % MFMA-busy is not an occupancy problem)
  const int r = 32 * wave + (lane & 31), g = lane >> 5;
  const bool rvalid = r < validB;
  float4 hr[4][4];
  HBPre<float, HB ? HB_NPRE_OFF : 1> hbp;
  auto prologue = [&]() __attribute__((always_inline)) {
#ifdef THX_EXP_NO_HBLOAD   // timing experiment (WRONG results): the off-diagonal tiles without their table / value loads
    if constexpr (HB) hbp.empty();
#else
    if constexpr (HB) hbp.load(hb, b, i, j, tid);
#endif
    if constexpr (!HB) {
      const float* Hrow = H + mat + (int64_t)(row0 + (rvalid ? r : 0)) * ld + col0 + 4 * g;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) hr[cb][q] = *reinterpret_cast<const float4*>(Hrow + 32 * cb + 8 * q);
    }
    if (!upd) {  // panel: global -> LDS (swizzled), one 16-byte piece of each of the ten sub-blocks per thread
      const float* Pn = panel + ((int64_t)b * ntiles + j) * TILE * TILE;
      const int pi = tid >> 3, pc = tid & 7;
      float* dst = Pc + pi * 32 + ((pc ^ ((pi >> 1) & 7)) << 2);
      const float* src = Pn + pi * TILE + 4 * pc;
      static_for<4>([&](auto is) __attribute__((always_inline)) {
        constexpr int sb = decltype(is)::value;
        static_for<sb + 1>([&](auto it) __attribute__((always_inline)) {
          constexpr int tb = decltype(it)::value;
          *reinterpret_cast<uint4*>(dst + (sb * (sb + 1) / 2 + tb) * 1024) =
              *reinterpret_cast<const uint4*>(src + 32 * sb * TILE + 32 * tb);
        });
      });
    }
  };
#ifdef THX_OFF_PROLOGUE_FIRST   // the round-1 order (A/B timing): panel + H loads and the panel copy BEFORE the first chunk's loads
  prologue();
#endif

  Engine<float>::Acc P;
  Engine<float>::zero(P);
#ifdef THX_OFF_PROF
  st[1] = (long long)__builtin_readcyclecounter();
#endif
  // (two LDS staging buffers with ONE barrier per k-chunk instead of one buffer with two -- panel copy moved behind the
  //  loop to keep 2 workgroups/CU -- measured the same 9.3-9.4 k cycles per chunk: the barriers are not the K-loop's limit)
  const float* Ap = L + lmat + (lf.packed ? 0 : (int64_t)col0 * ld) + kcol0;   // rows of block row j (operand A) / i (operand B)
  const float* Bp = L + lmat + (lf.packed ? 0 : (int64_t)row0 * ld) + kcol0;
  const int validA = upd ? tile_rows(pat, n, j) : TILE;   // (update: tile column k may be the LAST block row)
#ifdef THX_OFF_PROLOGUE_FIRST
  kloop<float, false>(Ap, validA, Bp, validB, ldt, Kspan, sA, sB, P, tid, nullptr, nullptr, NoHook{}, klist, ksa, ksb, lf.pstride);
#else
  kloop<float, false>(Ap, validA, Bp, validB, ldt, Kspan, sA, sB, P, tid, nullptr, nullptr, prologue, klist, ksa, ksb, lf.pstride);
#endif
#ifdef THX_OFF_PROF
  __builtin_amdgcn_sched_barrier(0);
  st[2] = (long long)__builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
#endif
  if constexpr (HB) {
    // block-compact H: the tile's pieces are gathered into the (now free) staging buffers, 64 rows at a time, and read back in
    // the accumulator layout -- a few hundred elements instead of a 64 KB tile of zeros from HBM
    constexpr int LDH = 132;
    static_assert(64 * LDH <= OFF32_STAGE_FLOATS, "half an H tile must fit in the staging buffers");
    __syncthreads();   // the K-loop's last chunk has been consumed
    // P = -sum first, H_ij's pieces are ADDED
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int q = 0; q < 16; ++q) P.v[cb][q] = -P.v[cb][q];
#ifdef THX_EXP_NO_HBLOAD
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int q = 0; q < 16; ++q) {   // (full-entropy mantissas: a constant pattern draws less power under the cap)
        const unsigned hsh = (unsigned)(tid * 64 + cb * 16 + q + 4099 * blockIdx.x) * 2654435761u;
        P.v[cb][q] += ((lane & 1) ? 1e-3f : -1e-3f) * (1.f + (float)(hsh >> 8) * (1.f / 16777216.f));
      }
#endif
    if constexpr (HB == HB_MODE_SCATTER) {
      // a few pieces per tile (pose graphs): added by the matrix cores, see hb_scatter.  (hb_add's barrier also publishes the
      // panel copy -- also when the K-loop had no iterations)
      hb_add<float, Engine<float>::Acc, decltype(hbp), 256 * HB_NPRE_OFF>(P, hbp, hb, b, smem, 0, hbp.cnt / (hb.bd * hb.bd), true, tid);
    } else {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      for (int k = tid; k < 64 * LDH / 4; k += 256) reinterpret_cast<float4*>(smem)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      __syncthreads();
      hbp.foreach(hb, b, tid, [&](int rr, int cc, float v) __attribute__((always_inline)) {
        if ((rr >> 6) == half) smem[(rr & 63) * LDH + cc] = v;
      });
      __syncthreads();
      if ((wave >> 1) == half) {
        const float* hrow = smem + (r & 63) * LDH + 4 * g;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 h = *reinterpret_cast<const float4*>(hrow + 32 * cb + 8 * q);
            P.v[cb][4 * q + 0] = h.x + P.v[cb][4 * q + 0];   // (P holds -sum already)
            P.v[cb][4 * q + 1] = h.y + P.v[cb][4 * q + 1];
            P.v[cb][4 * q + 2] = h.z + P.v[cb][4 * q + 2];
            P.v[cb][4 * q + 3] = h.w + P.v[cb][4 * q + 3];
          }
      }
      __syncthreads();
    }
    }
  } else {
    // P = H_ij - sum (rows outside the matrix: zero)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 h = hr[cb][q];
        P.v[cb][4 * q + 0] = (rvalid ? h.x : 0.f) - P.v[cb][4 * q + 0];
        P.v[cb][4 * q + 1] = (rvalid ? h.y : 0.f) - P.v[cb][4 * q + 1];
        P.v[cb][4 * q + 2] = (rvalid ? h.z : 0.f) - P.v[cb][4 * q + 2];
        P.v[cb][4 * q + 3] = (rvalid ? h.w : 0.f) - P.v[cb][4 * q + 3];
      }
    __syncthreads();  // panel copy visible (also when the K-loop had no iterations)
  }
#ifdef THX_OFF_PROF
  __builtin_amdgcn_sched_barrier(0);
  st[5] = (long long)__builtin_readcyclecounter();   // P = H - sum done (block-compact H: the gather rounds)
  __builtin_amdgcn_sched_barrier(0);
#endif
  if (upd) {   // trailing update: the tile goes back as it is (a diagonal tile: its lower triangle, zeros above)
    if (rvalid) {
      float* Lrow = Lij + (int64_t)r * ldt + 4 * g;
      const int rt = 32 * wave + (lane & 31);   // row inside the tile
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = 32 * cb + 8 * q + 4 * g;
          const bool dg = i == j;
          *reinterpret_cast<float4*>(Lrow + 32 * cb + 8 * q) =
              make_float4(dg && c + 0 > rt ? 0.f : P.v[cb][4 * q], dg && c + 1 > rt ? 0.f : P.v[cb][4 * q + 1],
                          dg && c + 2 > rt ? 0.f : P.v[cb][4 * q + 2], dg && c + 3 > rt ? 0.f : P.v[cb][4 * q + 3]);
        }
    }
    return;
  }
  Engine<float>::Acc X;
  Engine<float>::zero(X);
#ifdef THX_EXP_FULLINV   // timing experiment: the dataflow of a FULL 128 x 128 inverse in the panel, X_s = sum_{t <= s} W_st P_t --
                         // the same ten block products without the dependent chain (garbage results with today's panel)
  static_for<4>([&](auto is) __attribute__((always_inline)) {
    constexpr int sb = decltype(is)::value;
    static_for<sb + 1>([&](auto it) __attribute__((always_inline)) {
      constexpr int tb = decltype(it)::value;
      sub_mma_sw<sb, tb>(Pc, P, X, lane);
    });
  });
#else
  static_for<4>([&](auto is) __attribute__((always_inline)) {
    constexpr int sb = decltype(is)::value;
    static_for<sb>([&](auto it) __attribute__((always_inline)) {
      constexpr int tb = decltype(it)::value;
      sub_mma_sw<sb, tb>(Pc, X, P, lane);  // P_s += (-L_st) X_t
    });
    sub_mma_sw<sb, sb>(Pc, P, X, lane);    // X_s  = W_ss P_s
  });
#endif
#ifdef THX_OFF_PROF
  __builtin_amdgcn_sched_barrier(0);
  st[3] = (long long)__builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
#endif
  if (rvalid) {
    float* Lrow = Lij + (int64_t)r * ldt + 4 * g;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(Lrow + 32 * cb + 8 * q) =
            make_float4(X.v[cb][4 * q], X.v[cb][4 * q + 1], X.v[cb][4 * q + 2], X.v[cb][4 * q + 3]);
  }
  if (pat.rl_y) {   // right-looking forward substitution: block i of the vector loses L_ij y_j (a row's 128 columns sit in two lanes)
    float* yb = static_cast<float*>(pat.rl_y) + (int64_t)b * pat.rl_ldv;
    const float* yj = yb + col0 + 4 * g;
    float dot = 0.f;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 yv = *reinterpret_cast<const float4*>(yj + 32 * cb + 8 * q);
        dot += X.v[cb][4 * q] * yv.x + X.v[cb][4 * q + 1] * yv.y + X.v[cb][4 * q + 2] * yv.z + X.v[cb][4 * q + 3] * yv.w;
      }
    dot += __shfl_xor(dot, 32);
    if (g == 0 && rvalid) yb[row0 + r] -= dot;
  }
#ifdef THX_OFF_PROF
  __builtin_amdgcn_s_waitcnt(0);
  st[4] = (long long)__builtin_readcyclecounter();
  __syncthreads();
  if (tid == 0) {
    float* o = Lij;
    for (int k = 1; k < 5; ++k) o[k] = (float)(st[k] - st[0]);
    o[5] = (float)((long long)wall_clock64() - wc0);  // 100 MHz ticks
    o[6] = (float)(st[5] - st[0]);
    // where and when this workgroup ran (tools/prof/off_occupancy.py): raw 32-bit words
    o[7] = __int_as_float((int)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4));    // HW_REG_HW_ID
    o[8] = __int_as_float((int)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 20));   // HW_REG_XCC_ID
    o[9] = __int_as_float((int)(unsigned)wc0);
    o[10] = __int_as_float((int)(unsigned)wall_clock64());
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// chol_offdiag2, fp32, dense L frame: tiles (i, j) AND (i, j + 1) of row tile i >= j + 2 in one workgroup (round 5).
// Under thx_chol_factor the socket sits at its 1400 W cap (rocm-smi: 1356 W, 2.2 GHz; an HBM copy alone costs ~140 W per
// TB/s, profiles/r5/q_, r_): the factorisation's 2.7 TB/s are a quarter of its power.  Left-looking, tile (i, j) streams row
// panel L_i,0:j once per COLUMN j; here it is streamed once per column PAIR -- the K-loop over block columns 0 .. j - 1 stages
// three operand chunks (rows j, rows j + 1, rows i) for two tile products (4 for 2 before: -25 % operand loads, staging stores
// and barriers per MFMA, half the row-panel bytes from HBM), then
//   X0 = (H_ij - P0) L_jj^-T                      (the substitution of chol_offdiag, stored as L_ij)
//   P1 += X0 L_{j+1,j}^T                           (block column j's share of tile (i, j + 1): X0 stays in the accumulator
//                                                   registers and is the MFMAs' B operand itself -- an MFMA's k index is only a
//                                                   pairing of columns, and the accumulator layout pairs its columns the way the
//                                                   staged fragments do; the four chunks of L_{j+1,j} are staged as in the K-loop)
//   X1 = (H_i,j+1 - P1) L_{j+1,j+1}^-T
// Every accumulator receives the SAME MFMAs in the SAME order as in chol_offdiag_f32_kernel: the factor is bit-identical.
// Needs diag(j), tile (j + 1, j) and diag(j + 1) before it: the host launches column j's head tile alone (factor_impl).
// LDS 76 KB (two workgroups per CU): [0, 54 KB) three staging buffers | [36 KB, 76 KB) the panel copy of the substitution in
// progress (lands there straight from global memory after the K-loop; overlaps the third staging buffer only).
// ------------------------------------------------------------------------------------------------
constexpr int OFF2_PANEL_OFF = 2 * 128 * 36;              // floats: right behind staging buffers 0 and 1
constexpr int OFF2_SMEM = (OFF2_PANEL_OFF + 10 * 1024) * 4;   // 76 KB
static_assert(64 * 132 <= OFF2_PANEL_OFF, "the H gather (half a tile) must not touch the panel copy");
static_assert(2 * 128 * 36 <= OFF2_PANEL_OFF, "staging buffers 0 and 1 must not touch the panel copy");
static_assert(OFF2_SMEM >= 3 * 128 * 36 * 4 && 2 * OFF2_SMEM <= 160 * 1024, "three staging buffers; two workgroups per CU");

template <int HB>   // (as chol_offdiag_f32_kernel)
__global__ void __launch_bounds__(256, 2)
chol_offdiag2_f32_kernel(const float* __restrict__ H, float* __restrict__ L, const float* __restrict__ panel, int n,
                         int64_t ld, int j, int ntiles, int i_first, int nrow_tiles, int B, HBlk hb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int b = (slot / nrow_tiles) * 8 + xcd;   // problem-major: all row tiles of a problem on ONE XCD (panel rows j, j + 1 in its L2)
  const int i = i_first + slot % nrow_tiles;
  if (b >= B) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t mat = (int64_t)b * ld * ld;
  const int col0 = j * TILE, row0 = i * TILE;
  const int validB = min(TILE, n - row0);
  float* sA0 = smem;
  float* sA1 = smem + 128 * 36;
  float* sB = smem + 2 * 128 * 36;
  float* Pc = smem + OFF2_PANEL_OFF;
  const int r = 32 * wave + (lane & 31), g = lane >> 5, rl = lane & 31;
  const bool rvalid = r < validB;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

  HBPre2<float, HB ? 2 * HB_NPRE_OFF : 1> hb2;
  // ---- K-loop over block columns 0 .. j - 1: P0 += L_i L_j^T, P1 += L_i L_{j+1}^T ----
  const int lrow = tid >> 3, lc = tid & 7;
  unsigned voff[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) voff[u] = (unsigned)(((lrow + 32 * u) * (int)ld + lc * 4) * 4);
  const float* Lb = L + mat;
  const __amdgpu_buffer_rsrc_t rsA0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Lb + (int64_t)col0 * ld), 0, (int)(TILE * ld * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Lb + (int64_t)(col0 + TILE) * ld), 0, (int)(TILE * ld * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Lb + (int64_t)row0 * ld), 0, (int)(validB * ld * 4), 0x00020000);
  uint4 q0[4], q1[4], qb[4];
  auto gload3 = [&](int kc) __attribute__((always_inline)) {
    const int so = kc * 32 * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsA0, voff[u], so, 0);
      const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rsA1, voff[u], so, 0);
      const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rsB, voff[u], so, 0);
      q0[u] = make_uint4(a.x, a.y, a.z, a.w);
      q1[u] = make_uint4(c.x, c.y, c.z, c.w);
      qb[u] = make_uint4(d.x, d.y, d.z, d.w);
    }
  };
  auto gload1 = [&](int kc) __attribute__((always_inline)) {   // rows j + 1 only (block column j's share)
    const int so = kc * 32 * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rsA1, voff[u], so, 0);
      q1[u] = make_uint4(c.x, c.y, c.z, c.w);
    }
  };
  Engine<float>::Acc P0, P1;
  Engine<float>::zero(P0);
  Engine<float>::zero(P1);
  const int nk = 4 * j;
  if (nk > 0) gload3(0);
#ifdef THX_EXP_NO_HBLOAD
  if constexpr (HB) hb2.empty();
#else
  if constexpr (HB) hb2.load(hb, b, i, j, tid);
#endif
  const float* sBw = sB + 32 * wave * 36;
  for (int kc = 0; kc < nk; ++kc) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = lrow + 32 * u;
      *reinterpret_cast<uint4*>(sA0 + row * 36 + 4 * lc) = q0[u];
      *reinterpret_cast<uint4*>(sA1 + row * 36 + 4 * lc) = q1[u];
      *reinterpret_cast<uint4*>(sB + row * 36 + 4 * lc) = qb[u];
    }
    __syncthreads();
    if (kc + 1 < nk) gload3(kc + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float4 fb = *reinterpret_cast<const float4*>(sBw + rl * 36 + 8 * ks + 4 * g);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const float4 fa = *reinterpret_cast<const float4*>(sA0 + (32 * cb + rl) * 36 + 8 * ks + 4 * g);
        P0.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, P0.v[cb], 0, 0, 0);
        P0.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, P0.v[cb], 0, 0, 0);
        P0.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, P0.v[cb], 0, 0, 0);
        P0.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, P0.v[cb], 0, 0, 0);
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const float4 fa = *reinterpret_cast<const float4*>(sA1 + (32 * cb + rl) * 36 + 8 * ks + 4 * g);
        P1.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, P1.v[cb], 0, 0, 0);
        P1.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, P1.v[cb], 0, 0, 0);
        P1.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, P1.v[cb], 0, 0, 0);
        P1.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, P1.v[cb], 0, 0, 0);
      }
    }
  }

  // panel of diagonal tile jj: global -> LDS directly (global_load_lds: no registers -- ten pieces per thread held across the H
  // gather were spilled), in the substitution's swizzled layout: LDS slot (row pi, 16-byte slot ps) of a sub-block receives the
  // row's piece ps ^ ((pi >> 1) & 7); a wave fills 1 KB of consecutive slots per instruction
  auto panel_dma = [&](int jj) __attribute__((always_inline)) {
    const float* Pn = panel + ((int64_t)b * ntiles + jj) * TILE * TILE;
    const int pi = 8 * wave + (lane >> 3), pc = (lane & 7) ^ ((pi >> 1) & 7);
    const float* src = Pn + pi * TILE + 4 * pc;
    static_for<4>([&](auto is) __attribute__((always_inline)) {
      constexpr int sb = decltype(is)::value;
      static_for<sb + 1>([&](auto it) __attribute__((always_inline)) {
        constexpr int tb = decltype(it)::value;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 32 * sb * TILE + 32 * tb),
                                         (__attribute__((address_space(3))) void*)(Pc + (sb * (sb + 1) / 2 + tb) * 1024 + wave * 256),
                                         16, 0, 0);
      });
    });
  };
  // P <- H_(i, jj) - P  (block-compact H: the tile's pieces through the free staging buffers, 64 rows at a time; dense: loads)
  auto h_minus = [&](Engine<float>::Acc& P, int sel, int jj) __attribute__((always_inline)) {
    if constexpr (HB) {
      constexpr int LDH = 132;
      // P = -sum first, H's pieces are ADDED
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 16; ++q) P.v[cb][q] = -P.v[cb][q];
#ifdef THX_EXP_NO_HBLOAD
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const unsigned hsh = (unsigned)(tid * 64 + cb * 16 + q + 4099 * blockIdx.x + 77 * sel) * 2654435761u;
          P.v[cb][q] += ((lane & 1) ? 1e-3f : -1e-3f) * (1.f + (float)(hsh >> 8) * (1.f / 16777216.f));
        }
#endif
      if constexpr (HB == HB_MODE_SCATTER) {
        // a few pieces per tile (pose graphs): added by the matrix cores, see hb_scatter.  Both tiles' values go to staging
        // buffer 0 as ONE list before tile (i, j)'s pieces are applied (the caller's barrier before panel_dma(j): the K-loop is
        // done with the buffers); nothing writes that buffer until tile (i, j + 1)'s turn (column j's share of it is staged in
        // buffer 1): no second copy, no second barrier
        const int n0 = hb2.p1 - hb2.p0, ntot = hb2.cnt / (hb.bd * hb.bd);
        static_assert(256 * 2 * HB_NPRE_OFF + 64 * 36 <= 128 * 36, "list + overflow chunk inside staging buffer 0");
        hb_add<float, Engine<float>::Acc, decltype(hb2), 256 * 2 * HB_NPRE_OFF>(P, hb2, hb, b, smem, sel == 0 ? 0 : n0, sel == 0 ? n0 : ntot,
                                                                               sel == 0, tid);
      } else {
      __syncthreads();   // whatever read the staging buffers last is done
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        for (int k = tid; k < 64 * LDH / 4; k += 256) reinterpret_cast<float4*>(smem)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        hb2.foreach(hb, b, tid, sel, [&](int rr, int cc, float v) __attribute__((always_inline)) {
          if ((rr >> 6) == half) smem[(rr & 63) * LDH + cc] = v;
        });
        __syncthreads();
        if ((wave >> 1) == half) {
          const float* hrow = smem + (r & 63) * LDH + 4 * g;
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 h = *reinterpret_cast<const float4*>(hrow + 32 * cb + 8 * q);
              P.v[cb][4 * q + 0] = h.x + P.v[cb][4 * q + 0];   // (P holds -sum already)
              P.v[cb][4 * q + 1] = h.y + P.v[cb][4 * q + 1];
              P.v[cb][4 * q + 2] = h.z + P.v[cb][4 * q + 2];
              P.v[cb][4 * q + 3] = h.w + P.v[cb][4 * q + 3];
            }
        }
        __syncthreads();
      }
      }
    } else {
      const float* Hrow = H + mat + (int64_t)(row0 + (rvalid ? r : 0)) * ld + jj * TILE + 4 * g;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        float4 h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = *reinterpret_cast<const float4*>(Hrow + 32 * cb + 8 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          P.v[cb][4 * q + 0] = (rvalid ? h[q].x : 0.f) - P.v[cb][4 * q + 0];
          P.v[cb][4 * q + 1] = (rvalid ? h[q].y : 0.f) - P.v[cb][4 * q + 1];
          P.v[cb][4 * q + 2] = (rvalid ? h[q].z : 0.f) - P.v[cb][4 * q + 2];
          P.v[cb][4 * q + 3] = (rvalid ? h[q].w : 0.f) - P.v[cb][4 * q + 3];
        }
      }
    }
  };
  // X = P L_jj^-T with the panel copy in LDS (chol_offdiag's substitution), X -> tile (i, jj) of L
  auto substitute_store = [&](Engine<float>::Acc& P, Engine<float>::Acc& X, int jj) __attribute__((always_inline)) {
    Engine<float>::zero(X);
    static_for<4>([&](auto is) __attribute__((always_inline)) {
      constexpr int sb = decltype(is)::value;
      static_for<sb>([&](auto it) __attribute__((always_inline)) {
        constexpr int tb = decltype(it)::value;
        sub_mma_sw<sb, tb>(Pc, X, P, lane);  // P_s += (-L_st) X_t
      });
      sub_mma_sw<sb, sb>(Pc, P, X, lane);    // X_s  = W_ss P_s
    });
    if (rvalid) {
      float* Lrow = L + mat + (int64_t)(row0 + r) * ld + jj * TILE + 4 * g;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(Lrow + 32 * cb + 8 * q) =
              make_float4(X.v[cb][4 * q], X.v[cb][4 * q + 1], X.v[cb][4 * q + 2], X.v[cb][4 * q + 3]);
    }
  };

  // ---- tile (i, j) ----
  __syncthreads();      // the K-loop's last chunk has been consumed: the panel copy overlaps staging buffer 2
  panel_dma(j);
  h_minus(P0, 0, j);
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's pieces of the panel have landed
  __syncthreads();
  gload1(nk);           // first chunk of L_{j+1,j}: in flight under the substitution
  Engine<float>::Acc X0;
  substitute_store(P0, X0, j);
  // ---- block column j's share of tile (i, j + 1): P1 += X0 L_{j+1,j}^T, X0 from registers ----
  static_for<4>([&](auto ic) __attribute__((always_inline)) {
    constexpr int c = decltype(ic)::value;
    __syncthreads();    // (c = 0: the substitution's reads of the panel copy are done as well)
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4*>(sA1 + (lrow + 32 * u) * 36 + 4 * lc) = q1[u];
    __syncthreads();
    if (c == 0) panel_dma(j + 1);   // lands under the four chunks' MFMAs
    if (c < 3) gload1(nk + c + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const float4 fa = *reinterpret_cast<const float4*>(sA1 + (32 * cb + rl) * 36 + 8 * ks + 4 * g);
        P1.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, X0.v[c][4 * ks + 0], P1.v[cb], 0, 0, 0);
        P1.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, X0.v[c][4 * ks + 1], P1.v[cb], 0, 0, 0);
        P1.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, X0.v[c][4 * ks + 2], P1.v[cb], 0, 0, 0);
        P1.v[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, X0.v[c][4 * ks + 3], P1.v[cb], 0, 0, 0);
      }
    }
  });
  // ---- tile (i, j + 1) ----
  h_minus(P1, 1, j + 1);
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
  __syncthreads();
  substitute_store(P1, X0, j + 1);
}

// (A variant computing TWO row tiles per workgroup -- the column panel streamed once for both products, 3 staged operand
//  tiles per 2 tile products, half the workgroups, H / panel loaded after the K-loop, 240 VGPRs, 54 KB LDS -- measured
//  48.5 ms against 48.3 ms for this kernel on the same box at n = 1536, batch 4096, and the same at n = 3072 / batch 256 and
//  batch 1024: what it saves per tile in the prologue and the K-loop it gives back in the exposed loads and the longer
//  substitution phase.  Not kept; profiles/r2/f_pair_kernel_ab.txt.)
// ------------------------------------------------------------------------------------------------
// chol_offdiag, fp64: two workgroups per CU (a full 128x130 fp64 panel tile in LDS would be 133 KB):
//   * H_ij and the result go global <-> registers directly in the accumulator's native layout (a 4-lane group covers 32
//     contiguous bytes of a row);
//   * the panel's lower 32x32 sub-blocks are staged compactly (8 KB each, XOR-swizzled: conflict-free ds_read_b64 of the
//     A fragments) in two phases -- rows 0..2 (48 KB), then row 3 -- over the K-loop's staging buffers;
//   * the substitution runs IN PLACE: X_s = W_ss P_s goes through a 32-VGPR temporary back into P_s's registers, which
//     then serve as the B operand of the updates P_u += (-L_us) X_s.  128 + 32 accumulator VGPRs instead of 256.
// ------------------------------------------------------------------------------------------------
// LDS of the fp64 off-diagonal kernel (round 4): the K-loop's staging buffers (2 x 128 x LDT doubles = 36.9 KB with 16-column
// chunks; after the K-loop: four 8 KB panel sub-blocks) + a 40 KB region E for panel sub-blocks 0..4, which land there straight
// from global memory (global_load_lds, no registers) while the FIRST k-chunk is in flight -- 76.9 KB, two workgroups per CU.
// The substitution starts on E the moment the K-loop ends; sub-blocks 5..8 are requested then (LDS-direct into the staging buffers)
// and arrive under its first five block products, W_33 takes sub-block 0's place in E under the next four.  (Round 3: all ten
// sub-blocks were fetched after the K-loop, in two phases, each an exposed round trip.)
constexpr int OFF64_STAGE = (2 * 128 * CT<double>::LDT * 8 > 4 * 1024 * 8) ? 2 * 128 * CT<double>::LDT * 8 : 4 * 1024 * 8;
constexpr int OFF64_EBLK = 5;
constexpr int OFF64_SMEM = OFF64_STAGE + OFF64_EBLK * 1024 * 8;
static_assert(2 * OFF64_SMEM <= 160 * 1024, "two fp64 off-diagonal workgroups per CU");

// D.block(S) += Pc[block idx] * Bs.block(Tt)^T, Pc block: 32 x 32 doubles, element (r, c) at r * 32 + (c ^ 2 (r & 15))
template <int S, int Tt, typename DT>
__device__ __forceinline__ void sub_mma64(const double* blk, const Engine<double>::Acc& Bs, DT& D0, DT& D1, int lane) {
  // D0 / D1: the two 16-column accumulator blocks [h] of sub-block S: f64x4 (&)[2] each ([h])
  const int rl = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {       // 16-row half of the panel sub-block <-> accumulator block cp = 2S + ch
#pragma unroll
    for (int cbh = 0; cbh < 2; ++cbh)    // 16-column half <-> B block cb = 2Tt + cbh
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) {
        const int r = 16 * ch + rl, c = 16 * cbh + 4 * rho + kq;
        const double a = blk[r * 32 + (c ^ (2 * rl))];
        auto& d = ch == 0 ? D0 : D1;
        d[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs.v[0][2 * Tt + cbh][rho], d[0], 0, 0, 0);
        d[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs.v[1][2 * Tt + cbh][rho], d[1], 0, 0, 0);
      }
  }
}

template <int HB, bool RL = false>    // (HB: as chol_offdiag_f32_kernel; RL: the right-looking schedule's modes, TilePat.rl / rl_y -- an instance of its own: in the
                                      //  left-looking instance the extra live values spilled 268 - 700 B per lane through scratch)
__global__ void __launch_bounds__(256, 2)
chol_offdiag_f64_kernel(const double* __restrict__ H, double* __restrict__ L, const double* __restrict__ panel, int n,
                        int64_t ld, int jarg, int ntiles, int i_first, int nrow_tiles, int B, TilePat pat, HBlk hb) {
  using E = Engine<double>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* smem = reinterpret_cast<double*>(smem_raw);
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int b8 = gridDim.x / (8 * nrow_tiles);       // (the two block maps: chol_offdiag_f32_kernel)
  const int b = pat.lpt ? (slot % b8) * 8 + xcd : (slot / nrow_tiles) * 8 + xcd;
  const int rslot = pat.lpt ? slot / b8 : slot % nrow_tiles;
  // (tile-sparse: i_first = first ENTRY of the launch, relative to the column's list -- level schedule: absolute, and the entry
  //  names its block column)
  const int ent = pat.col_row ? (pat.ent_col ? 0 : pat.col_ptr[jarg]) + i_first + rslot : 0;
  // (right-looking schedule, TilePat.rl: see chol_offdiag_f32_kernel)
  const bool combo = RL && pat.rl_nsub > 0;
  const bool upd = RL && (combo ? rslot >= pat.rl_nsub : pat.rl >= 2);
  const int jc = combo ? jarg - 1 : pat.rl - 2;
  const int ub = combo ? jarg + 1 : jc + 1;
  const bool sla = RL && pat.rl == 1 && pat.rl_la != 0;
  int ui = 0, uk = 0;
  if (RL && upd) {   // (i_first: the launch's first slot, see chol_offdiag_f32_kernel)
    const int us = combo ? rslot - pat.rl_nsub : rslot + i_first;
    ui = (int)((__builtin_sqrtf(8.f * (float)us + 1.f) - 1.f) * 0.5f);
    while ((ui + 1) * (ui + 2) / 2 <= us) ++ui;
    while (ui * (ui + 1) / 2 > us) --ui;
    uk = us - ui * (ui + 1) / 2;
  }
  const int j = upd ? ub + uk : (pat.ent_col ? pat.ent_col[ent] : jarg);
  const int i = upd ? ub + ui : (pat.col_row ? pat.col_row[ent] : i_first + rslot);
  const int32_t* klist = pat.col_row ? pat.tile_k + pat.tile_kptr[ent] : nullptr;
  const int Kspan = (RL && pat.rl) ? ((upd || sla) ? TILE : 0) : (pat.col_row ? (pat.tile_kptr[ent + 1] - pat.tile_kptr[ent]) * TILE : j * TILE);
  const int kcol0 = upd ? jc * TILE : (sla ? (jarg - 1) * TILE : 0);
  if (b >= B) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int rl = lane & 15, kq = lane >> 4;
  const LFrame lf = lframe(pat, ld);
  const int64_t mat = (int64_t)b * ld * ld;            // H (dense frame)
  const int64_t lmat = (int64_t)b * lf.pstride;        // L (dense frame or tile-packed)
  const int64_t ldt = lf.ld;
  double* const Lij = L + lmat + lf.tile(i, j, ntiles + ent);
  const int32_t* ksa = lf.packed ? pat.tile_sa + pat.tile_kptr[ent] : nullptr;
  const int32_t* ksb = lf.packed ? pat.tile_sb + pat.tile_kptr[ent] : nullptr;
  const int col0 = j * TILE, row0 = i * TILE;
  const int validB = tile_rows(pat, n, i);
  double* sA = smem;
  double* sB = smem + 128 * CT<double>::LDT;

#ifdef THX_OFF_PROF64   // timing build (tools/prof/off_prof64.py): cycle stamps overwrite the head of the output tile
  long long st64[16];
  for (int k = 0; k < 16; ++k) st64[k] = 0;
  st64[0] = (long long)__builtin_readcyclecounter();
  const long long wc64 = (long long)wall_clock64();
#define THX_ST64(k) do { __builtin_amdgcn_sched_barrier(0); st64[k] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define THX_ST64(k)
#endif
  E::Acc P;
  E::zero(P);
  HBPre<double, HB ? HB_NPRE_OFF : 1> hbp;
#ifdef THX_EXP_NO_HBLOAD
  if constexpr (HB) hbp.empty();
#else
  if constexpr (HB) hbp.load(hb, b, i, j, tid);
#endif
  // panel sub-block q (row-major list of the lower triangle): block row SB[q], block column TB[q]
  const double* Pn = panel + ((int64_t)b * ntiles + j) * TILE * TILE;
  double* const smemE = smem + OFF64_STAGE / 8;
  // sub-blocks 0..4 -> E by LDS-direct loads: lane l of wave w, pass u writes the 16-byte unit U = 256 u + 64 w + l of the block
  // (row r = U / 16, unit u' = U % 16) and fetches the unit u' ^ (r & 15) of that row -- the XOR swizzle sub_mma64 reads with
  auto prefetch_panel = [&]() __attribute__((always_inline)) {
    if (upd) return;   // (trailing update: no substitution, no panel)
    constexpr int SB[5] = {0, 1, 1, 2, 2}, TB[5] = {0, 0, 1, 0, 1};
#pragma unroll
    for (int q = 0; q < OFF64_EBLK; ++q)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int U = 256 * u + 64 * wave + lane, r = U >> 4, up = U & 15;
        const double* src = Pn + (32 * SB[q] + r) * TILE + 32 * TB[q] + 2 * (up ^ (r & 15));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smemE + q * 1024 + (256 * u + 64 * wave) * 2),
                                         16, 0, 0);
      }
  };
  kloop<double, false>(L + lmat + (lf.packed ? 0 : (int64_t)col0 * ld) + kcol0, upd ? tile_rows(pat, n, j) : TILE,
                       L + lmat + (lf.packed ? 0 : (int64_t)row0 * ld) + kcol0, validB,
                       ldt, Kspan, sA, sB, P, tid, nullptr, nullptr, prefetch_panel, klist, ksa, ksb, lf.pstride);
  // sub-blocks 5..8 go LDS-direct into the staging buffers as soon as those are free (dense H: now, next to the H loads;
  // block-compact H: after the gather rounds), W_33 (sub-block 9) LDS-direct into sub-block 0's place in E once E has been read.
  // No panel data in registers: round 4 parked sub-blocks 5..9 (block-compact H) / W_33 (dense H) in VGPRs from here on and
  // hipcc spilled them -- 160 B per thread through scratch, 1 GB of extra HBM traffic per launch (profiles/r5/ab_, ac_).
  THX_ST64(1);   // K-loop done
  if (!HB && !upd) {
    // dense H: 5..8 straight into the staging buffers (LDS-direct, no registers: the 128 VGPRs of the H tile are about to be in
    // flight) -- after a barrier: the K-loop ends on a chunk's MFMAs, a slower wave may still be reading its fragments
    __syncthreads();
    constexpr int SB[4] = {2, 3, 3, 3}, TB[4] = {2, 0, 1, 2};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int U = 256 * u + 64 * wave + lane, r = U >> 4, up = U & 15;
        const double* src = Pn + (32 * SB[q] + r) * TILE + 32 * TB[q] + 2 * (up ^ (r & 15));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + q * 1024 + (256 * u + 64 * wave) * 2),
                                         16, 0, 0);
      }
  }
  // one panel sub-block (block row sbr, block column sbc) -> LDS at dst, LDS-direct, in sub_mma64's swizzled layout
  auto panel_dma = [&](int sbr, int sbc, double* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int U = 256 * u + 64 * wave + lane, r = U >> 4, up = U & 15;
      const double* src = Pn + (32 * sbr + r) * TILE + 32 * sbc + 2 * (up ^ (r & 15));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + (256 * u + 64 * wave) * 2), 16, 0, 0);
    }
  };
  if constexpr (HB) {
    // block-compact H: the tile's pieces through the (free) staging buffers, 32 rows -- one wave's -- at a time
    constexpr int LDH = 130;
    static_assert(32 * LDH * 8 <= OFF64_STAGE, "a quarter of an H tile must fit in the staging buffers");
    __syncthreads();
    THX_ST64(8);    // (sub-stamps 8..11: the first barrier -- the K-loop's last MFMAs drained --, then the gather rounds 1..3 begin)
    // P = -sum first, H_ij's pieces are ADDED
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) P.v[h][cb][rho] = -P.v[h][cb][rho];
#ifdef THX_EXP_NO_HBLOAD
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) {
          const unsigned hsh = (unsigned)(tid * 64 + h * 32 + cb * 4 + rho + 4099 * blockIdx.x) * 2654435761u;
          const unsigned h2 = hsh * 2246822519u + 1u;
          P.v[h][cb][rho] += ((lane & 1) ? 1e-3 : -1e-3) * (1.0 + ((double)hsh * 4294967296.0 + (double)h2) * (1.0 / 18446744073709551616.0));
        }
#endif
    if constexpr (HB == HB_MODE_SCATTER) {
      // a few pieces per tile (pose graphs): added by the matrix cores, see hb_scatter
      static_assert((256 * HB_NPRE_OFF + 64 * 36) * 8 <= OFF64_STAGE, "list + overflow chunk inside the staging buffers");
      hb_add<double, E::Acc, decltype(hbp), 256 * HB_NPRE_OFF>(P, hbp, hb, b, smem, 0, hbp.cnt / (hb.bd * hb.bd), true, tid);
#ifndef THX_EXP_NO_HBBARRIER   // (timing experiment, WRONG results)
      __syncthreads();   // the list has been read: panel sub-blocks 5..8 may take the staging buffers
#endif
    } else {
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      if (rd == 1) THX_ST64(9);
      if (rd == 2) THX_ST64(10);
      if (rd == 3) THX_ST64(11);
      for (int k = tid; k < 32 * LDH / 2; k += 256) reinterpret_cast<double2*>(smem)[k] = make_double2(0.0, 0.0);
      __syncthreads();
      hbp.foreach(hb, b, tid, [&](int rr, int cc, double v) __attribute__((always_inline)) {
        if ((rr >> 5) == rd) smem[(rr & 31) * LDH + cc] = v;
      });
      __syncthreads();
      if (wave == rd) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double* hrow = smem + (16 * h + rl) * LDH + kq;
#pragma unroll
          for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int rho = 0; rho < 4; ++rho) P.v[h][cb][rho] = hrow[16 * cb + 4 * rho] + P.v[h][cb][rho];   // (P holds -sum already)
        }
      }
      __syncthreads();
    }
    }
  } else {
  // ---- P = H_ij - sum, H straight from global memory in the native layout (rows outside the matrix: zero) ----
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = 32 * wave + 16 * h + rl;
    const bool rv = r < validB;
    const double* Hrow = H + mat + (int64_t)(row0 + (rv ? r : 0)) * ld + col0 + kq;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) {
        const double hv = Hrow[16 * cb + 4 * rho];
        P.v[h][cb][rho] = (rv ? hv : 0.0) - P.v[h][cb][rho];
      }
  }
  }
  THX_ST64(2);   // P = H - sum
  if (RL && upd) {   // trailing update: the tile goes back as it is (a diagonal tile: its lower triangle, zeros above)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = 32 * wave + 16 * h + rl;
      if (r < validB) {
        double* Lrow = Lij + (int64_t)r * ldt + kq;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
          for (int rho = 0; rho < 4; ++rho) {
            const int c = 16 * cb + 4 * rho + kq;
            Lrow[16 * cb + 4 * rho] = (i == j && c > r) ? 0.0 : P.v[h][cb][rho];
          }
      }
    }
    return;
  }
  if constexpr (HB) {
    // sub-blocks 5..8 -> the staging buffers, LDS-direct, now that the gather rounds are done with them (their last barrier has
    // passed); they land under the first five block products, which read E.  E itself was requested in the prologue: every wave
    // has waited for its own pieces inside the K-loop (older loads) -- or right here when the K-loop was empty -- and the gather's
    // barriers published them.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (Kspan == 0) __syncthreads();
    panel_dma(2, 2, smem + 0 * 1024);
    panel_dma(3, 0, smem + 1 * 1024);
    panel_dma(3, 1, smem + 2 * 1024);
    panel_dma(3, 2, smem + 3 * 1024);
  } else {
    // E (LDS-direct loads of the prologue) and the staging buffers complete and visible to every wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  THX_ST64(3);   // panel staged and visible
  // ---- in-place substitution ----
  auto solve_diag = [&](auto is, const double* Wss) __attribute__((always_inline)) {
    constexpr int sb = decltype(is)::value;
    f64x4 T0[2], T1[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) { T0[h][r4] = 0.0; T1[h][r4] = 0.0; }
    sub_mma64<sb, sb>(Wss, P, T0, T1, lane);  // X_s = W_ss P_s
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      P.v[h][2 * sb] = T0[h];
      P.v[h][2 * sb + 1] = T1[h];
    }
  };
  auto update = [&](auto is, auto it, const double* Mst) __attribute__((always_inline)) {
    constexpr int sb = decltype(is)::value, tb = decltype(it)::value;
    f64x4 D0[2] = {P.v[0][2 * sb], P.v[1][2 * sb]}, D1[2] = {P.v[0][2 * sb + 1], P.v[1][2 * sb + 1]};
    sub_mma64<sb, tb>(Mst, P, D0, D1, lane);  // P_s += (-L_st) X_t  (X_t lives in P_t's registers)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      P.v[h][2 * sb] = D0[h];
      P.v[h][2 * sb + 1] = D1[h];
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  // sub-blocks 0..4 from E (there since the first k-chunk)
  solve_diag(I0{}, smemE + 0 * 1024);
  update(I1{}, I0{}, smemE + 1 * 1024);
  solve_diag(I1{}, smemE + 2 * 1024);
  update(I2{}, I0{}, smemE + 3 * 1024);
  update(I2{}, I1{}, smemE + 4 * 1024);
  if constexpr (HB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // sub-blocks 5..8 (LDS-direct) and W_33 have landed
  __syncthreads();                  // every wave is done with E (block-compact H: and sees sub-blocks 5..8)
  panel_dma(3, 3, smemE + 0 * 1024);   // sub-block 9 = W_33 takes sub-block 0's place: LDS-direct, lands under the next four block
                                       // products (round 4 parked it in 8 VGPRs from the K-loop's end on: spilled to scratch)
  solve_diag(I2{}, smem + 0 * 1024);
  update(I3{}, I0{}, smem + 1 * 1024);
  update(I3{}, I1{}, smem + 2 * 1024);
  update(I3{}, I2{}, smem + 3 * 1024);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                  // W_33 in place
  solve_diag(I3{}, smemE + 0 * 1024);
  THX_ST64(4);   // substitution done
  // ---- store X (in P's registers) ----
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = 32 * wave + 16 * h + rl;
    if (r < validB) {
      double* Lrow = Lij + (int64_t)r * ldt + kq;
#pragma unroll
      for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) Lrow[16 * cb + 4 * rho] = P.v[h][cb][rho];
    }
  }
  if (RL && pat.rl_y) {   // right-looking forward substitution: block i of the vector loses L_ij y_j (a row's 128 columns sit in four lanes)
    double* yb = static_cast<double*>(pat.rl_y) + (int64_t)b * pat.rl_ldv;
    const double* yj = yb + col0 + kq;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double dot = 0.0;
#pragma unroll
      for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) dot += P.v[h][cb][rho] * yj[16 * cb + 4 * rho];
      dot += __shfl_xor(dot, 16);
      dot += __shfl_xor(dot, 32);
      const int r = 32 * wave + 16 * h + rl;
      if (kq == 0 && r < validB) yb[row0 + r] -= dot;
    }
  }
#ifdef THX_OFF_PROF64
  __builtin_amdgcn_s_waitcnt(0);
  st64[5] = (long long)__builtin_readcyclecounter();
  __syncthreads();
  if (tid == 0) {
    double* o = Lij;
    for (int k = 1; k < 6; ++k) o[k] = (double)(st64[k] - st64[0]);
    o[6] = (double)((long long)wall_clock64() - wc64);  // 100 MHz ticks
    for (int k = 8; k < 12; ++k) o[k] = (double)(st64[k] - st64[0]);
  }
#endif
#undef THX_ST64
}

// ------------------------------------------------------------------------------------------------
// chol_offdiag, fp64, EIGHT waves per workgroup (round 6): the same tile, staging, panel plan and arithmetic -- element by element
// the same sequence of MFMAs, bit-identical results -- with a wave owning 16 rows x 128 columns instead of 32 x 128: 64 accumulator
// VGPRs, <= 128 in all, two workgroups = FOUR waves per SIMD.  Why: the 4-wave kernel is two waves per SIMD (256 VGPRs); with
// K-loops of zero to three tiles -- block columns 0 ... 3, 27 of the 95 ms -- a tile's time is its serial epilogue (H pieces, ten
// dependent block products of the substitution, 128 KB of stores), which one other wave per SIMD cannot cover: 0.39 ... 0.71 of
// the peak per executed flop (profiles/r5/x_).  Half the epilogue per wave and twice the waves to interleave.  MEASURED
// (profiles/r6/ae_): the early columns gain NOTHING (their tiles are serial phases -- pieces, panel waits, ten dependent block
// products, stores -- that more waves of the SAME tile do not overlap; it takes more TILES per CU), the late ones 0.1 - 0.3 ms
// each: 91.5 -> 90.4 ms with every column on this kernel, which is the default.  Left-looking column schedule only (no TilePat.rl
// modes, no tile pattern); block-compact H through the matrix-core scatter or a dense H frame (HB_MODE_ROUNDS: the 4-wave kernel).
// ------------------------------------------------------------------------------------------------
template <int S, int Tt>
__device__ __forceinline__ void sub_mma64_16(const double* blk, const Acc16& Bs, f64x4& D0, f64x4& D1, int lane) {
  const int rl = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ch = 0; ch < 2; ++ch)
#pragma unroll
    for (int cbh = 0; cbh < 2; ++cbh)
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) {
        const int r = 16 * ch + rl, c = 16 * cbh + 4 * rho + kq;
        const double a = blk[r * 32 + (c ^ (2 * rl))];
        auto& d = ch == 0 ? D0 : D1;
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs.v[2 * Tt + cbh][rho], d, 0, 0, 0);
      }
}

template <int HB>
__global__ void __launch_bounds__(512, 4)   // (second argument: waves per SIMD -- two workgroups of eight per CU)
chol_offdiag_f64w8_kernel(const double* __restrict__ H, double* __restrict__ L, const double* __restrict__ panel, int n,
                          int64_t ld, int jarg, int ntiles, int i_first, int nrow_tiles, int B, TilePat pat, HBlk hb) {
  static_assert(HB != HB_MODE_ROUNDS, "dense tiles of H: the 4-wave kernel");
  constexpr int NT = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* smem = reinterpret_cast<double*>(smem_raw);
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int b8 = gridDim.x / (8 * nrow_tiles);       // (the two block maps: chol_offdiag_f32_kernel)
  const int b = pat.lpt ? (slot % b8) * 8 + xcd : (slot / nrow_tiles) * 8 + xcd;
  const int rslot = pat.lpt ? slot / b8 : slot % nrow_tiles;
  const int ent = pat.col_row ? (pat.ent_col ? 0 : pat.col_ptr[jarg]) + i_first + rslot : 0;
  const int j = pat.ent_col ? pat.ent_col[ent] : jarg;
  const int i = pat.col_row ? pat.col_row[ent] : i_first + rslot;
  const int32_t* klist = pat.col_row ? pat.tile_k + pat.tile_kptr[ent] : nullptr;
  const int Kspan = pat.col_row ? (pat.tile_kptr[ent + 1] - pat.tile_kptr[ent]) * TILE : j * TILE;
  if (b >= B) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int rl = lane & 15, kq = lane >> 4;
  const LFrame lf = lframe(pat, ld);
  const int64_t mat = (int64_t)b * ld * ld;
  const int64_t lmat = (int64_t)b * lf.pstride;
  const int64_t ldt = lf.ld;
  double* const Lij = L + lmat + lf.tile(i, j, ntiles + ent);
  const int32_t* ksa = lf.packed ? pat.tile_sa + pat.tile_kptr[ent] : nullptr;
  const int32_t* ksb = lf.packed ? pat.tile_sb + pat.tile_kptr[ent] : nullptr;
  const int col0 = j * TILE, row0 = i * TILE;
  const int validB = tile_rows(pat, n, i);
  double* sA = smem;
  double* sB = smem + 128 * CT<double>::LDT;
  Acc16 P;
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
#pragma unroll
    for (int k = 0; k < 4; ++k) P.v[cb][k] = 0.0;
  HBPre<double, HB ? 2 : 1, NT> hbp;
  if constexpr (HB != 0) hbp.load(hb, b, i, j, tid);
  const double* Pn = panel + ((int64_t)b * ntiles + j) * TILE * TILE;
  double* const smemE = smem + OFF64_STAGE / 8;
  // one panel sub-block (block row sbr, block column sbc) -> LDS at dst, LDS-direct, in sub_mma64's swizzled layout: thread U
  // writes the 16-byte unit U of the block (row r = U / 16, unit U % 16) and fetches the unit (U % 16) ^ (r & 15) of that row
  auto panel_dma = [&](int sbr, int sbc, double* dst) __attribute__((always_inline)) {
    const int U = tid, r = U >> 4, up = U & 15;
    const double* src = Pn + (32 * sbr + r) * TILE + 32 * sbc + 2 * (up ^ (r & 15));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(dst + (64 * wave) * 2), 16, 0, 0);
  };
  auto prefetch_panel = [&]() __attribute__((always_inline)) {   // sub-blocks 0..4 -> E (chol_offdiag_f64_kernel)
    panel_dma(0, 0, smemE + 0 * 1024);
    panel_dma(1, 0, smemE + 1 * 1024);
    panel_dma(1, 1, smemE + 2 * 1024);
    panel_dma(2, 0, smemE + 3 * 1024);
    panel_dma(2, 1, smemE + 4 * 1024);
  };
  {
    const double* sBw = sB + 16 * wave * CT<double>::LDT;
    kloop_f<double, false, false, CT<double>::LDT, false, NT>(
        L + lmat + (lf.packed ? 0 : (int64_t)col0 * ld), TILE, L + lmat + (lf.packed ? 0 : (int64_t)row0 * ld), validB, ldt, Kspan, sA,
        sB, tid, nullptr, nullptr,
        [&]() __attribute__((always_inline)) {
          constexpr int LDT = CT<double>::LDT;
#pragma unroll
          for (int ks = 0; ks < CT<double>::KB / 8; ++ks) {
            const double2 fb = *reinterpret_cast<const double2*>(sBw + rl * LDT + 8 * ks + 2 * kq);
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
              const double2 fa = *reinterpret_cast<const double2*>(sA + (16 * cb + rl) * LDT + 8 * ks + 2 * kq);
              P.v[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa.x, fb.x, P.v[cb], 0, 0, 0);
              P.v[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa.y, fb.y, P.v[cb], 0, 0, 0);
            }
          }
        },
        prefetch_panel, klist, ksa, ksb, lf.pstride);
  }
  const int r = 16 * wave + rl;   // this lane's tile row
  if constexpr (HB == 0) {
    // dense H: sub-blocks 5..8 straight into the staging buffers (after a barrier: a slower wave may still read its fragments)
    __syncthreads();
    panel_dma(2, 2, smem + 0 * 1024);
    panel_dma(3, 0, smem + 1 * 1024);
    panel_dma(3, 1, smem + 2 * 1024);
    panel_dma(3, 2, smem + 3 * 1024);
    const bool rv = r < validB;
    const double* Hrow = H + mat + (int64_t)(row0 + (rv ? r : 0)) * ld + col0 + kq;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) {
        const double hv = Hrow[16 * cb + 4 * rho];
        P.v[cb][rho] = (rv ? hv : 0.0) - P.v[cb][rho];
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // E and the staging buffers complete and visible to every wave
  } else {
    __syncthreads();   // the K-loop's last chunk has been consumed: the staging buffers are free
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) P.v[cb][rho] = -P.v[cb][rho];
    static_assert((NT * 2 + 64 * 36) * 8 <= OFF64_STAGE, "list + overflow chunk inside the staging buffers");
    hb_add<double, Acc16, decltype(hbp), NT * 2, NT>(P, hbp, hb, b, smem, 0, hbp.cnt / (hb.bd * hb.bd), true, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of E (requested in the prologue) have landed
    __syncthreads();   // the list has been read; E visible to every wave (also when the K-loop was empty)
    panel_dma(2, 2, smem + 0 * 1024);   // sub-blocks 5..8: they land under the first five block products, which read E
    panel_dma(3, 0, smem + 1 * 1024);
    panel_dma(3, 1, smem + 2 * 1024);
    panel_dma(3, 2, smem + 3 * 1024);
  }
  // ---- in-place substitution (chol_offdiag_f64_kernel's, on 16 rows) ----
  auto solve_diag = [&](auto is, const double* Wss) __attribute__((always_inline)) {
    constexpr int sb = decltype(is)::value;
    f64x4 T0, T1;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) { T0[r4] = 0.0; T1[r4] = 0.0; }
    sub_mma64_16<sb, sb>(Wss, P, T0, T1, lane);  // X_s = W_ss P_s
    P.v[2 * sb] = T0;
    P.v[2 * sb + 1] = T1;
  };
  auto update = [&](auto is, auto it, const double* Mst) __attribute__((always_inline)) {
    constexpr int sb = decltype(is)::value, tb = decltype(it)::value;
    sub_mma64_16<sb, tb>(Mst, P, P.v[2 * sb], P.v[2 * sb + 1], lane);  // P_s += (-L_st) X_t
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  solve_diag(I0{}, smemE + 0 * 1024);
  update(I1{}, I0{}, smemE + 1 * 1024);
  solve_diag(I1{}, smemE + 2 * 1024);
  update(I2{}, I0{}, smemE + 3 * 1024);
  update(I2{}, I1{}, smemE + 4 * 1024);
  if constexpr (HB != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // sub-blocks 5..8 have landed
  __syncthreads();                       // every wave is done with E (block-compact H: and sees sub-blocks 5..8)
  panel_dma(3, 3, smemE + 0 * 1024);     // W_33 takes sub-block 0's place, lands under the next four block products
  solve_diag(I2{}, smem + 0 * 1024);
  update(I3{}, I0{}, smem + 1 * 1024);
  update(I3{}, I1{}, smem + 2 * 1024);
  update(I3{}, I2{}, smem + 3 * 1024);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                       // W_33 in place
  solve_diag(I3{}, smemE + 0 * 1024);
  // ---- store X ----
  if (r < validB) {
    double* Lrow = Lij + (int64_t)r * ldt + kq;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) Lrow[16 * cb + 4 * rho] = P.v[cb][rho];
  }
}

// ------------------------------------------------------------------------------------------------
// chol_offdiag, fp64, HALF TILES (round 6): a workgroup of four waves produces 64 rows x 128 columns of the tile (16 rows
// per wave, Acc16), with 36.8 KB of LDS -- the K-loop's staging buffers and nothing else -- and <= 128 VGPRs: FOUR workgroups per CU
// instead of two.  For the block columns with K-loops of zero to three tiles, whose tiles are chains of latency-bound phases (pieces
// of H, panel waits, ten dependent block products, stores) that two resident workgroups cannot overlap.  The price: each half
// stages the whole column panel L_j (1.5x the operand traffic per tile product), the solve panel is not prefetched under the K-loop
// but fetched afterwards, four sub-blocks at a time into the free staging buffers (three exposed round trips), one k-chunk in
// flight instead of two.  Same MFMAs in the same order per element: bit-identical (tools/cmp_f64_half.py, tests/test_gpu_block_hessian.py).
// MEASURED (profiles/r6/af_): n = 1536, batch 4096: 90.0 -> 88.5 ms with the first 6 - 8 block columns on this kernel (0.700 -> 0.711),
// every further column gives 0.1 ms back (the K-loop with one chunk in flight and 1.5x the staging loses to the 8-wave kernel from
// ~8 tiles on): thx_chol_schedule.f64_half_max_ktiles, default 8.
// ------------------------------------------------------------------------------------------------
#ifndef THX_F64H_AHEAD
#define THX_F64H_AHEAD 1   // k-chunks in flight (2: 142 VGPRs wanted, spills -- see the header comment)
#endif
template <int HB>
__global__ void __launch_bounds__(256, 4)
chol_offdiag_f64h_kernel(const double* __restrict__ H, double* __restrict__ L, const double* __restrict__ panel, int n,
                         int64_t ld, int jarg, int ntiles, int i_first, int nrow_tiles, int B, TilePat pat, HBlk hb) {
  static_assert(HB != HB_MODE_ROUNDS, "dense tiles of H: the 4-wave full-tile kernel");
  constexpr int NT = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* smem = reinterpret_cast<double*>(smem_raw);
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nslots = 2 * nrow_tiles;                 // (two halves per row tile, adjacent slots)
  const int b = (slot / nslots) * 8 + xcd;
  const int hslot = slot % nslots;
  const int half = hslot & 1, rslot = hslot >> 1;
  const int j = jarg, i = i_first + rslot;
  const int Kspan = j * TILE;
  if (b >= B) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int rl = lane & 15, kq = lane >> 4;
  const int64_t mat = (int64_t)b * ld * ld;
  const int col0 = j * TILE, row0 = i * TILE + 64 * half;
  const int validB = min(64, tile_rows(pat, n, i) - 64 * half);   // rows of this half inside the matrix
  if (validB <= 0) return;
  double* const Lij = L + mat + (int64_t)row0 * ld + col0;
  double* sA = smem;
  double* sB = smem + 128 * CT<double>::LDT;
  Acc16 P;
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
#pragma unroll
    for (int k = 0; k < 4; ++k) P.v[cb][k] = 0.0;
  HBPre<double, HB ? HB_NPRE_OFF : 1, NT> hbp;
  if constexpr (HB != 0) hbp.load(hb, b, i, j, tid);
  const double* Pn = panel + ((int64_t)b * ntiles + j) * TILE * TILE;
  {
    const double* sBw = sB + 16 * wave * CT<double>::LDT;
    kloop_f<double, false, false, CT<double>::LDT, false, NT, THX_F64H_AHEAD, 64>(
        L + mat + (int64_t)col0 * ld, TILE, L + mat + (int64_t)row0 * ld, validB, ld, Kspan, sA, sB, tid, nullptr, nullptr,
        [&]() __attribute__((always_inline)) {
          constexpr int LDT = CT<double>::LDT;
#pragma unroll
          for (int ks = 0; ks < CT<double>::KB / 8; ++ks) {
            const double2 fb = *reinterpret_cast<const double2*>(sBw + rl * LDT + 8 * ks + 2 * kq);
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
              const double2 fa = *reinterpret_cast<const double2*>(sA + (16 * cb + rl) * LDT + 8 * ks + 2 * kq);
              P.v[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa.x, fb.x, P.v[cb], 0, 0, 0);
              P.v[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa.y, fb.y, P.v[cb], 0, 0, 0);
            }
          }
        });
  }
  const int r = 16 * wave + rl;   // this lane's row inside the half
  __syncthreads();                // the K-loop's last chunk has been consumed: the staging buffers are free
  if constexpr (HB == 0) {
    const bool rv = r < validB;
    const double* Hrow = H + mat + (int64_t)(row0 + (rv ? r : 0)) * ld + col0 + kq;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) {
        const double hv = Hrow[16 * cb + 4 * rho];
        P.v[cb][rho] = (rv ? hv : 0.0) - P.v[cb][rho];
      }
  } else {
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) P.v[cb][rho] = -P.v[cb][rho];
    static_assert((NT * HB_NPRE_OFF + 64 * 36) * 8 <= OFF64_STAGE, "list + overflow chunk inside the staging buffers");
    // (hb_scatter's "wave" names the 16-row block of the TILE: 4 half + wave)
    {
      const int bd = hb.bd, bb = bd * bd;
      const int nreg = min(min(hbp.cnt / bb, NT * HB_NPRE_OFF / bb), 64), np = hbp.cnt / bb;
      const int wv = __builtin_amdgcn_readfirstlane(4 * half + wave);
      hbp.to_list(smem, tid);
      __syncthreads();
      hb_scatter(P, smem, hbp.wmeta, 0, nreg, bd, wv, lane);
      if (np > nreg) {   // (workgroup uniform) crowded tile: further chunks of 64 pieces from memory (hb_add)
        const double* base = static_cast<const double*>(hb.blocks) + (int64_t)b * hb.bstride;
        double* over = smem + NT * HB_NPRE_OFF;
        for (int q0 = nreg; q0 < np; q0 += 64) {
          const int nq = min(64, np - q0);
          __syncthreads();
          for (int idx = tid; idx < nq * bb; idx += NT) over[idx] = base[(int64_t)hb.piece_blk[hbp.p0 + q0 + idx / bb] * bb + idx % bb];
          const int wm = hb.piece_rc[hbp.p0 + q0 + min(lane, nq - 1)];
          __syncthreads();
          hb_scatter(P, over, wm, 0, nq, bd, wv, lane);
        }
      }
    }
    __syncthreads();   // the list has been read: the panel may take the staging buffers
  }
  // one panel sub-block -> LDS slot, LDS-direct, in sub_mma64's swizzled layout (two passes of the 256 threads)
  auto panel_dma = [&](int sbr, int sbc, double* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int U = 256 * u + tid, rr = U >> 4, up = U & 15;
      const double* src = Pn + (32 * sbr + rr) * TILE + 32 * sbc + 2 * (up ^ (rr & 15));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + (256 * u + 64 * wave) * 2), 16, 0, 0);
    }
  };
  auto landed = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  auto solve_diag = [&](auto is, const double* Wss) __attribute__((always_inline)) {
    constexpr int sb = decltype(is)::value;
    f64x4 T0, T1;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) { T0[r4] = 0.0; T1[r4] = 0.0; }
    sub_mma64_16<sb, sb>(Wss, P, T0, T1, lane);
    P.v[2 * sb] = T0;
    P.v[2 * sb + 1] = T1;
  };
  auto update = [&](auto is, auto it, const double* Mst) __attribute__((always_inline)) {
    constexpr int sb = decltype(is)::value, tb = decltype(it)::value;
    sub_mma64_16<sb, tb>(Mst, P, P.v[2 * sb], P.v[2 * sb + 1], lane);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
#ifndef THX_F64H_RING
  // ---- the substitution in three panel phases of four / four / two sub-blocks through the staging buffers (each an exposed round
  //      trip, covered by the other three workgroups of the CU) ----
  panel_dma(0, 0, smem + 0 * 1024);
  panel_dma(1, 0, smem + 1 * 1024);
  panel_dma(1, 1, smem + 2 * 1024);
  panel_dma(2, 0, smem + 3 * 1024);
  landed();
  solve_diag(I0{}, smem + 0 * 1024);
  update(I1{}, I0{}, smem + 1 * 1024);
  solve_diag(I1{}, smem + 2 * 1024);
  update(I2{}, I0{}, smem + 3 * 1024);
  __syncthreads();   // every wave is done with the four slots
  panel_dma(2, 1, smem + 0 * 1024);
  panel_dma(2, 2, smem + 1 * 1024);
  panel_dma(3, 0, smem + 2 * 1024);
  panel_dma(3, 1, smem + 3 * 1024);
  landed();
  update(I2{}, I1{}, smem + 0 * 1024);
  solve_diag(I2{}, smem + 1 * 1024);
  update(I3{}, I0{}, smem + 2 * 1024);
  update(I3{}, I1{}, smem + 3 * 1024);
  __syncthreads();
  panel_dma(3, 2, smem + 0 * 1024);
  panel_dma(3, 3, smem + 1 * 1024);
  landed();
  update(I3{}, I2{}, smem + 0 * 1024);
  solve_diag(I3{}, smem + 1 * 1024);
#else
  // (-DTHX_F64H_RING, measured EQUAL: 87.99 / 88.03 against 87.92 / 88.13 ms, profiles/r6/af_ -- not the default: ten barriers for nothing)
  // ---- the substitution with the panel's ten sub-blocks through a RING of four slots in the staging buffers: sub-block m lives in
  //      slot m % 4; the first four are requested here, sub-block k + 3 at step k >= 1 -- right behind the barrier that proves every
  //      wave has finished product k - 1, the previous tenant of that slot -- so a request has three products' time to land.  The
  //      waits count instructions: loads return in order, two LDS-direct loads per thread and sub-block, and behind sub-block k
  //      at most k + 1, k + 2 are in flight.  One exposed round trip instead of three. ----
  panel_dma(0, 0, smem + 0 * 1024);
  panel_dma(1, 0, smem + 1 * 1024);
  panel_dma(1, 1, smem + 2 * 1024);
  panel_dma(2, 0, smem + 3 * 1024);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __syncthreads();
  solve_diag(I0{}, smem + 0 * 1024);                 // product 0: (0,0)
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  panel_dma(2, 1, smem + 0 * 1024);                  // sub-block 4 -> slot 0
  update(I1{}, I0{}, smem + 1 * 1024);               // product 1: (1,0)
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  panel_dma(2, 2, smem + 1 * 1024);                  // 5 -> slot 1
  solve_diag(I1{}, smem + 2 * 1024);                 // product 2: (1,1)
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  panel_dma(3, 0, smem + 2 * 1024);                  // 6 -> slot 2
  update(I2{}, I0{}, smem + 3 * 1024);               // product 3: (2,0)
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  panel_dma(3, 1, smem + 3 * 1024);                  // 7 -> slot 3
  update(I2{}, I1{}, smem + 0 * 1024);               // product 4: (2,1)
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  panel_dma(3, 2, smem + 0 * 1024);                  // 8 -> slot 0
  solve_diag(I2{}, smem + 1 * 1024);                 // product 5: (2,2)
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  panel_dma(3, 3, smem + 1 * 1024);                  // 9 -> slot 1
  update(I3{}, I0{}, smem + 2 * 1024);               // product 6: (3,0)
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  update(I3{}, I1{}, smem + 3 * 1024);               // product 7: (3,1)
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  __syncthreads();
  update(I3{}, I2{}, smem + 0 * 1024);               // product 8: (3,2)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  solve_diag(I3{}, smem + 1 * 1024);                 // product 9: (3,3)
#endif
  // ---- store X ----
  if (r < validB) {
    double* Lrow = Lij + (int64_t)r * ld + kq;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) Lrow[16 * cb + 4 * rho] = P.v[cb][rho];
  }
}

// ------------------------------------------------------------------------------------------------
// triangular solves with one right-hand side per problem, one workgroup per problem, HBM bound
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T>
__device__ __forceinline__ void panel_g2l(const T* __restrict__ Pn, T* tile, int tid) {
  using C = CT<T>;
  constexpr int CPR = TILE / C::VEC, RPP = 256 / CPR;
  const int c = (tid % CPR) * C::VEC;
#pragma unroll 4
  for (int rr = tid / CPR; rr < TILE; rr += RPP)
    *reinterpret_cast<uint4*>(tile + rr * C::LDM + c) = *reinterpret_cast<const uint4*>(Pn + rr * TILE + c);
}

template <typename T>
static size_t solve_smem(int npad) {
  return (size_t)128 * CT<T>::LDM * sizeof(T) + (size_t)(npad + 128 + 32) * sizeof(T);
}

// Row-wise tile pattern of L for the list-driven solves (thx_chol_solve_sparse): for block row i the column tiles j < i with
// L_ij structurally non-zero.  Null pointers = dense.
struct RowPat {
  const int32_t* __restrict__ row_ptr;   // [ntiles + 1]
  const int32_t* __restrict__ row_tile;  // [row_ptr[ntiles]]
  const int32_t* __restrict__ row_slot;  // tile-packed factor: the slot of every listed tile (else nullptr)
  int32_t nslots;                        // tile-packed factor: slots per problem (else 0)
  // LEVEL schedule (thx_chol_solve_levels): one launch = the block rows [j0, j0 + gridDim.y) of one elimination-tree level (they do
  // not depend on each other), one workgroup per (problem, block row); j0 < 0: one workgroup walks all block rows of its problem
  int32_t j0;
  const int32_t* __restrict__ tile_valid;   // per-tile padding (see TilePat)
};

// L y = rhs (stand-alone; the LM iteration gets y from the factorisation).
// LIST = false: the whole solution vector lives in LDS (n <= ~23 k fp32 / 3.4 k fp64) and every tile of a block row is streamed.
// LIST = true : the tiles of the row's list only, and the vector stays in global memory (L2) -- no limit on n.
template <typename T, bool LIST>
__global__ void __launch_bounds__(256)
chol_fwd_kernel(const T* __restrict__ L, const T* __restrict__ panel, const T* __restrict__ rhs, T* __restrict__ y,
                int n, int64_t ld, int64_t ldv, int ntiles, RowPat rp) {
  using C = CT<T>;
  using V = typename C::V;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);
  const int npad = LIST ? 0 : ntiles * TILE;
  T* yv = tile + 128 * C::LDM;  // [npad]
  T* tv = yv + npad;            // [128]
  T* ubuf = tv + 128;           // [32]
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool packed = LIST && rp.nslots > 0;
  const T* Lb = L + (int64_t)b * (packed ? (int64_t)rp.nslots * TILE * TILE : ld * ld);
  const T* rb = rhs + (int64_t)b * ldv;
  T* yb = y + (int64_t)b * ldv;
  if constexpr (!LIST) {
    for (int k = tid; k < npad; k += 256) yv[k] = k < n ? rb[k] : T(0);
    __syncthreads();
  }
  constexpr int QPT = TILE / C::VEC;   // VEC-wide column groups per tile
  const int jlo = (LIST && rp.j0 >= 0) ? rp.j0 + (int)blockIdx.y : 0, jhi = (LIST && rp.j0 >= 0) ? jlo + 1 : ntiles;
  for (int jb = jlo; jb < jhi; ++jb) {
    const int row0 = jb * TILE, valid = (LIST && rp.tile_valid) ? rp.tile_valid[jb] : min(TILE, n - row0);
    panel_g2l<T>(panel + ((int64_t)b * ntiles + jb) * TILE * TILE, tile, tid);
    // t[r] = sum_{k < row0} L[row0 + r][k] y[k]: wave w takes rows r = w (mod 4), four rows in flight
    const int l0 = LIST ? rp.row_ptr[jb] : 0;
    const int items = LIST ? (rp.row_ptr[jb + 1] - l0) * QPT : row0 / C::VEC;
    for (int rr = wave; rr < TILE; rr += 16) {
      T s[4] = {T(0), T(0), T(0), T(0)};
      for (int it = lane; it < items; it += 64) {
        const int k = LIST ? rp.row_tile[l0 + it / QPT] * TILE + (it % QPT) * C::VEC : it * C::VEC;
        // the listed tile's rows: dense frame (row0 + r) * ld + k, or the packed tile's own 128 x 128 block
        const T* Lt_ = packed ? Lb + (int64_t)rp.row_slot[l0 + it / QPT] * TILE * TILE + (it % QPT) * C::VEC : Lb + (int64_t)row0 * ld + k;
        const int64_t lds_ = packed ? TILE : ld;
        V yk;
        if constexpr (LIST) {   // (scalar loads: a row of the vector need not be 16-byte aligned)
          if constexpr (sizeof(T) == 4) yk = V{yb[k], yb[k + 1], yb[k + 2], yb[k + 3]};
          else yk = V{yb[k], yb[k + 1]};
        } else {
          yk = *reinterpret_cast<const V*>(yv + k);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = rr + 4 * u;
          if (r < valid) {
            const V lv = *reinterpret_cast<const V*>(Lt_ + (int64_t)r * lds_);
            if constexpr (sizeof(T) == 4) s[u] += lv.x * yk.x + lv.y * yk.y + lv.z * yk.z + lv.w * yk.w;
            else s[u] += lv.x * yk.x + lv.y * yk.y;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const T t = wave_sum(s[u]);
        if (lane == 0) tv[rr + 4 * u] = t;
      }
    }
    __syncthreads();
    if (tid < TILE) {
      if constexpr (LIST) tv[tid] = (tid < valid ? rb[row0 + tid] : T(0)) - tv[tid];
      else tv[tid] = yv[row0 + tid] - tv[tid];
    }
    __syncthreads();
    if (wave == 0) panel_forward<T>(tile, tv, ubuf, lane);
    __syncthreads();
    if (tid < TILE) {
      if constexpr (LIST) {
        if (tid < valid) yb[row0 + tid] = tv[tid];
      } else {
        yv[row0 + tid] = tv[tid];
      }
    }
    __syncthreads();   // (LIST: also makes the block of y visible to the whole workgroup before the next row reads it)
  }
  if constexpr (!LIST)
    for (int k = tid; k < n; k += 256) yb[k] = yv[k];
}

// L^T x = y : right-looking from the last block row; every (LIST: every structurally non-zero) tile of L is streamed once
template <typename T, bool LIST>
__global__ void __launch_bounds__(256)
chol_bwd_kernel(const T* __restrict__ L, const T* __restrict__ panel, const T* __restrict__ yin, T* __restrict__ x,
                int n, int64_t ld, int64_t ldv, int ntiles, RowPat rp) {
  using C = CT<T>;
  using V = typename C::V;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);
  const int npad = LIST ? 0 : ntiles * TILE;
  T* z = tile + 128 * C::LDM;  // [npad]
  T* xb = z + npad;            // [128] current block
  T* ubuf = xb + 128;          // [32]
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool packed = LIST && rp.nslots > 0;
  const T* Lb = L + (int64_t)b * (packed ? (int64_t)rp.nslots * TILE * TILE : ld * ld);
  T* zg = x + (int64_t)b * ldv;   // LIST: the working vector IS the output (global, L2 resident)
  if constexpr (LIST) {
    if (yin != x && rp.j0 < 0)   // (level schedule: the host copies y into x before the first level's launch)
      for (int k = tid; k < n; k += 256) zg[k] = yin[(int64_t)b * ldv + k];
  } else {
    for (int k = tid; k < npad; k += 256) z[k] = k < n ? yin[(int64_t)b * ldv + k] : T(0);
  }
  __syncthreads();
  constexpr int QPT = TILE / C::VEC;
  // (level schedule: the rows of a level scatter into DISJOINT column blocks of z -- the non-zero rows of a block column are a
  //  chain of the elimination tree, no two of them on one level -- so the push needs no atomics)
  const int jhi = (LIST && rp.j0 >= 0) ? rp.j0 + (int)blockIdx.y : ntiles - 1, jlo = (LIST && rp.j0 >= 0) ? jhi : 0;
  for (int jb = jhi; jb >= jlo; --jb) {
    const int row0 = jb * TILE, valid = (LIST && rp.tile_valid) ? rp.tile_valid[jb] : min(TILE, n - row0);
    panel_g2l<T>(panel + ((int64_t)b * ntiles + jb) * TILE * TILE, tile, tid);
    if (tid < TILE) xb[tid] = LIST ? (tid < valid ? zg[row0 + tid] : T(0)) : z[row0 + tid];
    __syncthreads();
    if (wave == 0) panel_backward<T>(tile, xb, ubuf, lane);
    __syncthreads();
    if (tid < TILE) {
      if constexpr (LIST) {
        if (tid < valid) zg[row0 + tid] = xb[tid];
      } else {
        z[row0 + tid] = xb[tid];
      }
    }
    // z[0:row0] -= L[row0 : row0 + valid, 0:row0]^T x_block : a thread owns VEC consecutive columns,
    // rows unrolled by 8 (independent 16-byte loads in flight), x broadcast from LDS
    const int l0 = LIST ? rp.row_ptr[jb] : 0;
    const int items = LIST ? (rp.row_ptr[jb + 1] - l0) * QPT : row0 / C::VEC;
    for (int it = tid; it < items; it += 256) {
      const int k = LIST ? rp.row_tile[l0 + it / QPT] * TILE + (it % QPT) * C::VEC : it * C::VEC;
      T s[4] = {T(0), T(0), T(0), T(0)};
      const T* Lk = packed ? Lb + (int64_t)rp.row_slot[l0 + it / QPT] * TILE * TILE + (it % QPT) * C::VEC : Lb + (int64_t)row0 * ld + k;
      const int64_t lds_ = packed ? (int64_t)TILE : ld;   // (row stride of the listed tile)
      int rr = 0;
      for (; rr + 8 <= valid; rr += 8) {
        V lv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) lv[u] = *reinterpret_cast<const V*>(Lk + (int64_t)(rr + u) * lds_);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const T xv = xb[rr + u];
          // (explicit fused multiply-adds: -ffp-contract leaves the choice to the vectoriser, which mixes packed multiplies +
          //  adds into the chain -- the block-row kernel below must reproduce this sum bit for bit)
          if constexpr (sizeof(T) == 4) {
            s[0] = fma_t(lv[u].x, xv, s[0]); s[1] = fma_t(lv[u].y, xv, s[1]); s[2] = fma_t(lv[u].z, xv, s[2]); s[3] = fma_t(lv[u].w, xv, s[3]);
          } else {
            s[0] = fma_t(lv[u].x, xv, s[0]); s[1] = fma_t(lv[u].y, xv, s[1]);
          }
        }
      }
      for (; rr < valid; ++rr) {
        const V lv = *reinterpret_cast<const V*>(Lk + (int64_t)rr * lds_);
        const T xv = xb[rr];
        if constexpr (sizeof(T) == 4) {
          s[0] = fma_t(lv.x, xv, s[0]); s[1] = fma_t(lv.y, xv, s[1]); s[2] = fma_t(lv.z, xv, s[2]); s[3] = fma_t(lv.w, xv, s[3]);
        } else {
          s[0] = fma_t(lv.x, xv, s[0]); s[1] = fma_t(lv.y, xv, s[1]);
        }
      }
      if constexpr (LIST) {   // (scalar accesses: a row of the vector need not be 16-byte aligned, ldv = n = 6 P)
#pragma unroll
        for (int u = 0; u < C::VEC; ++u) zg[k + u] -= s[u];
      } else {
#pragma unroll
        for (int u = 0; u < C::VEC; ++u) z[k + u] -= s[u];
      }
    }
    __syncthreads();
  }
  if constexpr (!LIST)
    for (int k = tid; k < n; k += 256) x[(int64_t)b * ldv + k] = z[k];
}

// L^T x = y for SMALL batches on dense frames, one launch per block row instead of one workgroup per problem (round 6):
// chol_bwd_kernel streams a problem's whole lower triangle through ONE workgroup.  Here block row jb is its own launch, x is the
// working vector (z) in place:
//   workgroup (k, b), k < jb:  z_k -= L_jb,k^T x_jb   (x_jb is final: the previous launch finished it) -- and the workgroup of
//   k = jb - 1 then holds the finished z_jb-1 (rows above jb pushed into it in earlier launches: stream order) and turns it into
//   x_jb-1 = L_jj^-T z_jb-1 through the panel, in place: the only writer of that block in this launch.
// The first launch (jb = ntiles) only finishes the last block.  Per column the same sums in the same order as chol_bwd_kernel
// (rows ascending, fused multiply-adds; the same panel substitution): the solution is bit-identical.
template <typename T>
__global__ void __launch_bounds__(256)
chol_bwd_rows_kernel(const T* __restrict__ L, const T* __restrict__ panel, T* __restrict__ x, int n, int64_t ld, int64_t ldv,
                     int ntiles, int jb) {
  using C = CT<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);
  T* xb = tile + 128 * C::LDM;   // [128]
  T* ubuf = xb + 128;            // [32]
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  T* xg = x + (int64_t)b * ldv;
  if (jb < ntiles) {   // push of block row jb into column block k
    const int row0 = jb * TILE, valid = min(TILE, n - row0);
    if (tid < TILE) xb[tid] = tid < valid ? xg[row0 + tid] : T(0);
    __syncthreads();
    if (tid < TILE) {
      const T* Lk = L + (int64_t)b * ld * ld + (int64_t)row0 * ld + k * TILE + tid;
      T s = T(0);
      int rr = 0;
      for (; rr + 16 <= valid; rr += 16) {
        T lv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) lv[u] = Lk[(int64_t)(rr + u) * ld];
        // (explicit fused multiply-adds: left to -ffp-contract the vectoriser emits packed multiplies + separate adds for half of
        //  the chain -- other roundings than chol_bwd_kernel's v_fmac chain)
#pragma unroll
        for (int u = 0; u < 16; ++u) s = fma_t(lv[u], xb[rr + u], s);
      }
      for (; rr < valid; ++rr) s = fma_t(Lk[(int64_t)rr * ld], xb[rr], s);
      xg[k * TILE + tid] -= s;
    }
    if (k != jb - 1) return;   // (workgroup uniform)
    __syncthreads();           // block jb - 1 of x is finished and visible to this workgroup (its own writes)
  } else if (k != 0) {
    return;
  }
  // finish block jf = jb - 1: x_jf = L_jf,jf^-T z_jf through the panel
  const int jf = jb - 1, row0 = jf * TILE, valid = min(TILE, n - row0);
  panel_g2l<T>(panel + ((int64_t)b * ntiles + jf) * TILE * TILE, tile, tid);
  if (tid < TILE) xb[tid] = tid < valid ? xg[row0 + tid] : T(0);
  __syncthreads();
  if (wave == 0) panel_backward<T>(tile, xb, ubuf, lane);
  __syncthreads();
  if (tid < valid) xg[row0 + tid] = xb[tid];
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

// right-looking schedule: the damping of the diagonal elements d >= d0 of the working matrix in the L frame,
// A_dd += ellipsoidal ? lambda H_dd + eps : lambda, with H_dd the ORIGINAL diagonal (dense H frame or block list) -- block column 0
// gets its damping from chol_diag as always; the other diagonal tiles have just been written by the first trailing update
template <typename T>
__global__ void rl_damp_kernel(T* __restrict__ L, int64_t ld, const T* __restrict__ H, int64_t ldh, HBlk hb,
                               const T* __restrict__ damping, int ellipsoidal, T eps, int d0, int n) {
  const int b = blockIdx.y, d = d0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n) return;
  const T lam = damping[b];
  T add = lam;
  if (ellipsoidal) {
    T h;
    if (hb.blocks) {
      const int v = d / hb.bd, e = d % hb.bd;
      h = static_cast<const T*>(hb.blocks)[(int64_t)b * hb.bstride + (int64_t)hb.diag_blk[v] * hb.bd * hb.bd + e * hb.bd + e];
    } else {
      h = H[(int64_t)b * ldh * ldh + (int64_t)d * ldh + d];
    }
    add = lam * h + eps;
  }
  L[(int64_t)b * ld * ld + (int64_t)d * ld + d] += add;
}

// dst[b][k] = idx[k] >= 0 ? src[b][idx[k]] : 0  -- the solver's permuted / padded vectors <-> the linearization's (thx_vec_gather)
template <typename T>
__global__ void vec_gather_kernel(const T* __restrict__ src, int64_t lds, T* __restrict__ dst, int64_t ldd,
                                  const int32_t* __restrict__ idx, int n) {
  const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int i = idx[k];
  dst[(int64_t)b * ldd + k] = i >= 0 ? src[(int64_t)b * lds + i] : T(0);
}

constexpr size_t LDS_LIMIT = 160 * 1024;
constexpr int FACTOR_NEEDS_FORWARD = 1000;   // factor_impl -> factor_then_forward: factor done, y = L^-1 rhs still to be computed

// Defaults of the per-call schedule (thx_chol_schedule; a negative field / a NULL pointer selects them): read once from the
// environment, never written afterwards -- the library keeps no mutable schedule state, two callers with different schedules in
// one process do not see each other.
static const int g_split_diag_min_default = [] {
  const char* e = getenv("THX_CHOL_SPLIT_DIAG_MIN");
  return e ? atoi(e) : 2048;
}();
static const int g_column_pairs_default = [] {
  const char* e = getenv("THX_CHOL_COLPAIR");
  return e ? atoi(e) : 1;
}();
static const int g_right_looking_max_default = [] {
  const char* e = getenv("THX_CHOL_RL_MAX_BATCH");
  return e ? atoi(e) : -1;   // (-1: by dtype and size, factor_impl)
}();

// Launch-side state is kept PER DEVICE (a process may drive several GPUs, from several threads): the dynamic-LDS limits
// raised with hipFuncSetAttribute, and the auxiliary stream + events of the two-stream schedule, which belong to the device
// they were created on.
constexpr int MAX_DEVICES = 64;
struct DeviceLaunchState {
  size_t attr_diag[2][2] = {{0, 0}, {0, 0}}, attr_syrk[2][2] = {{0, 0}, {0, 0}}, attr_solve[2] = {0, 0}, attr_bwd_rows[2] = {0, 0};   // [0] float, [1] double (x HB)
  bool attr_off = false;
  hipStream_t aux[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_lag[2] = {nullptr, nullptr}, ev_join[2] = {nullptr, nullptr};
  hipEvent_t ev_diag = nullptr, ev_rest = nullptr;   // look-ahead schedule (one part): diag(j) done / rest of column j done
};
static std::mutex g_launch_mutex;
static DeviceLaunchState& launch_state() {
  static DeviceLaunchState st[MAX_DEVICES];
  int dev = 0;
  hipGetDevice(&dev);
  return st[(dev >= 0 && dev < MAX_DEVICES) ? dev : 0];
}

template <typename T>
static int factor_impl(const void* H, int64_t ld, int n, int B, const void* damping, int ellipsoidal, double eps,
                       void* L, void* panel, int32_t* info, const void* rhs, void* y, int64_t ldv, hipStream_t st,
                       const thx_tile_pattern* tp = nullptr, const HBlk* hbp = nullptr, const thx_level_schedule* ls = nullptr,
                       const thx_chol_schedule* sched = nullptr) {
  const bool use_hb = hbp != nullptr;
  const HBlk hb = use_hb ? *hbp : HBlk{nullptr, 0, 0, nullptr, nullptr, nullptr};
  // how an off-diagonal tile takes its pieces of H: a few per tile (pose graphs) -> added by the matrix cores (hb_scatter); many (a
  // bundle adjustment's reduced camera system: up to 21 x 21 blocks per tile) or unknown -> gathered through LDS in rounds
  static const int hb_scatter_max_default = [] {
    const char* e = getenv("THX_HB_SCATTER_MAX_PIECES");   // (0: always the gather rounds)
    return e ? atoi(e) : 64;
  }();
  const int hb_scatter_max = (sched && sched->hb_scatter_max_pieces >= 0) ? sched->hb_scatter_max_pieces : hb_scatter_max_default;
  const bool hb_sc = use_hb && hb.bd <= 6 && hb.max_tile_pieces > 0 && hb.max_tile_pieces <= hb_scatter_max;
  // fp64: the first f64_wide_max block columns (K-loops shorter than that many tiles) take the 8-wave off-diagonal kernel
  // (chol_offdiag_f64w8_kernel).  Default: ALL of them -- measured +1.0 ... 1.3 % at batch 256 / 1024 / 4096 (n = 1536), growing with
  // the number of columns that use it (profiles/r6/ae_): four waves per SIMD serve the K-loop better than two; the early columns,
  // for which the kernel was written, gain nothing.
  static const int f64_wide_default = [] {
    const char* e = getenv("THX_F64_WIDE_MAX_KTILES");   // (0: never)
    return e ? atoi(e) : (1 << 30);
  }();
  const int f64_wide_max = (sched && sched->f64_wide_max_ktiles >= 0) ? sched->f64_wide_max_ktiles : f64_wide_default;
  // ... and the first f64_half_max of them (K-loops shorter than that many tiles) the HALF-TILE kernel (chol_offdiag_f64h_kernel, four
  // workgroups per CU): measured optimum 6 ... 8 at n = 1536 / batch 4096 (profiles/r6/af_: 90.0 -> 88.5 ms; all twelve columns 89.0)
  static const int f64_half_default = [] {
    const char* e = getenv("THX_F64_HALF_MAX_KTILES");   // (0: never)
    return e ? atoi(e) : 8;
  }();
  const int f64_half_max = (sched && sched->f64_half_max_ktiles >= 0) ? sched->f64_half_max_ktiles : f64_half_default;
  const int ntiles = (n + TILE - 1) / TILE;
  TilePat pat{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr};
  // ld == 0: L is the TILE-PACKED factor (B, nslots, TILE, TILE) of the pattern
  const bool packed = ld == 0;
  if (packed && (!tp || tp->nslots <= 0 || !tp->tile_sa || !tp->tile_sb || !tp->diag_s))
    return fail("thx_chol_factor: a tile-packed factor (ld = 0) needs a tile pattern with slot tables");
  if (tp) {
    if (tp->ntiles != ntiles) return fail("thx_chol_factor_sparse: the tile pattern was built for another matrix order");
    pat = TilePat{tp->col_ptr, tp->col_row, tp->tile_kptr, tp->tile_k, tp->diag_kptr, tp->diag_k,
                  packed ? tp->tile_sa : nullptr, packed ? tp->tile_sb : nullptr, packed ? tp->diag_s : nullptr,
                  packed ? tp->nslots : 0, 0, ls ? ls->ent_col : nullptr, ls ? ls->tile_valid : nullptr};
  }
  const int64_t hstride = (int64_t)ld * ld;                                            // H frame (dense H only; never packed)
  const int64_t lstride = packed ? (int64_t)tp->nslots * TILE * TILE : (int64_t)ld * ld;   // elements of L per problem
  if (packed && lstride * (int64_t)sizeof(T) > 0x7fffffffLL)
    return fail("thx_chol_factor: tile-packed factor larger than 2 GB per problem");
  // diagonal phase: chol_syrk_kernel + chol_potrf_kernel from split_diag_min problems per call on (measured, n = 1536: fp32
  // 45.1 vs 46.0 ms at batch 4096, fp64 101.6 vs 105.2 ms; equal at batch 1024; 3.74 vs 3.51 ms at batch 256 -- the second
  // launch per column costs more than the chain there), else the fused chol_diag_kernel
  const int split_diag_min = (sched && sched->split_diag_min_batch >= 0) ? sched->split_diag_min_batch : g_split_diag_min_default;
  const int column_pairs = (sched && sched->column_pairs >= 0) ? sched->column_pairs : g_column_pairs_default;
  // (default hand-over to the left-looking schedule, measured at 12 block columns with two launches per column, profiles/r6/ar_:
  //  fp32 right-looking wins through 64 problems -- batch 40 1.49 -> 1.16 ms, 64 1.66 -> 1.58 --, fp64 through 40; the update
  //  launches grow with the SQUARE of the block columns, so the limit shrinks with them, down to round 6's first 32)
  const int rl_auto = sizeof(T) == 4 ? min(64, max(32, 768 / max(ntiles, 1))) : min(40, max(32, 480 / max(ntiles, 1)));
  const int rl_max_batch = (sched && sched->right_looking_max_batch >= 0) ? sched->right_looking_max_batch
                           : (g_right_looking_max_default >= 0 ? g_right_looking_max_default : rl_auto);
  const bool fused_diag = B < split_diag_min;
  const size_t dsm = fused_diag ? DiagSmem<T>::bytes(rhs ? ntiles * TILE : 0) : SyrkSmem<T>::bytes(rhs ? ntiles * TILE : 0);
  if (dsm > LDS_LIMIT) return fail("thx_chol_factor: n too large for the fused forward substitution (LDS)");
  if (ls && (!packed || !use_hb)) return fail("thx_chol_factor_levels: tile-packed factor + block-compact H");
  // level schedule + fused forward substitution: a column keeps only its K-list's blocks of y in LDS -- the launch of level l is
  // sized for the longest K-list of that level (level_maxk_host)
  size_t ls_smem_max = 0;
  if (ls) {
    for (int l = 0; l < ls->nlevels; ++l) {
      const int yp = rhs ? ls->level_maxk_host[l] * TILE : 0;
      ls_smem_max = std::max(ls_smem_max, std::max(DiagSmem<T>::bytes(yp), SyrkSmem<T>::bytes(yp)));
    }
    if (ls_smem_max > LDS_LIMIT) return fail("thx_chol_factor_levels: K-list too long for the fused forward substitution (LDS)");
  }
  std::lock_guard<std::mutex> guard(g_launch_mutex);   // (the whole enqueue: the auxiliary stream / events are shared)
  DeviceLaunchState& ds = launch_state();
  constexpr int ti = sizeof(T) == 8;
  if (ls) {   // (both diagonal schedules may be taken, level by level; no y buffer: well below the default limit, raised anyway)
    const size_t d0 = ls_smem_max, s0 = ls_smem_max;
    if (d0 > ds.attr_diag[ti][1]) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(chol_diag_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)d0);
      ds.attr_diag[ti][1] = d0;
    }
    if (s0 > ds.attr_syrk[ti][1]) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(chol_syrk_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s0);
      ds.attr_syrk[ti][1] = s0;
    }
  }
  if (fused_diag && dsm > ds.attr_diag[ti][use_hb]) {
    hipFuncSetAttribute(use_hb ? reinterpret_cast<const void*>(chol_diag_kernel<T, true>)
                               : reinterpret_cast<const void*>(chol_diag_kernel<T, false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)dsm);
    ds.attr_diag[ti][use_hb] = dsm;
  }
  if (!fused_diag && dsm > ds.attr_syrk[ti][use_hb]) {
    hipFuncSetAttribute(use_hb ? reinterpret_cast<const void*>(chol_syrk_kernel<T, true>)
                               : reinterpret_cast<const void*>(chol_syrk_kernel<T, false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)dsm);
    ds.attr_syrk[ti][use_hb] = dsm;
  }
  if (!ds.attr_off) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f32_kernel<0>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF32_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f32_kernel<1>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF32_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f32_kernel<2>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF32_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag2_f32_kernel<0>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF2_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag2_f32_kernel<1>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF2_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag2_f32_kernel<2>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF2_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f64_kernel<0>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF64_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f64_kernel<1>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF64_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f64_kernel<2>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF64_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f64_kernel<0, true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF64_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f64_kernel<1, true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF64_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f64_kernel<2, true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF64_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f64h_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        OFF64_STAGE);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f64h_kernel<HB_MODE_SCATTER>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF64_STAGE);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f64w8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        OFF64_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_offdiag_f64w8_kernel<HB_MODE_SCATTER>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, OFF64_SMEM);
    ds.attr_off = true;
  }
  hipMemsetAsync(info, 0, sizeof(int32_t) * (size_t)B, st);
  // One half of the batch per stream, the second half one diagonal phase behind the first: the latency-bound serial part
  // of chol_diag (one wave per workgroup busy, §4.1) and the tail of every launch of one half run underneath the MFMA-bound
  // chol_offdiag of the other half.  The halves touch disjoint memory; the auxiliary stream forks from / joins the
  // caller's stream through events.  (Overlapping diag(j+1) with the rest of column j of the SAME problems had gained
  // nothing: column j+1 needs all of column j, so there is no slack to fill.)
  struct Half {
    hipStream_t s;
    int b0, nb;
  };
  static const int split_min = [] {
    const char* e = getenv("THX_CHOL_SPLIT_MIN");  // batch size from which the multi-stream schedule is used (0: never)
    return e ? atoi(e) : 1024;
  }();
  // number of parts (streams): 2; THX_CHOL_PARTS=3 staggers three thirds (measured)
  static const int nparts_cfg = [] {
    const char* e = getenv("THX_CHOL_PARTS");
    const int v = e ? atoi(e) : 2;
    return v < 2 ? 2 : (v > 3 ? 3 : v);
  }();
  const bool split = split_min > 0 && B >= split_min && ntiles > 1;
  const int nparts = split ? nparts_cfg : 1;
  Half halves[3] = {{st, 0, B}, {st, 0, 0}, {st, 0, 0}};
  if (split) {
    if (!ds.ev_fork) hipEventCreateWithFlags(&ds.ev_fork, hipEventDisableTiming);
    for (int k = 0; k + 1 < nparts; ++k)
      if (!ds.aux[k]) {
        hipStreamCreateWithFlags(&ds.aux[k], hipStreamNonBlocking);
        hipEventCreateWithFlags(&ds.ev_lag[k], hipEventDisableTiming);
        hipEventCreateWithFlags(&ds.ev_join[k], hipEventDisableTiming);
      }
    const int per = min((B / nparts + 7) / 8 * 8, B);  // (multiple of 8: the XCD-aware block map of chol_offdiag)
    int b0 = 0;
    for (int k = 0; k < nparts; ++k) {
      const int nb = k + 1 < nparts ? min(per, B - b0) : B - b0;
      halves[k] = {k == 0 ? st : ds.aux[k - 1], b0, nb};
      b0 += nb;
    }
    hipEventRecord(ds.ev_fork, st);
    for (int k = 0; k + 1 < nparts; ++k) hipStreamWaitEvent(ds.aux[k], ds.ev_fork, 0);
  }
  // (block-compact H: the half's problems start at h.b0 of the block list; H itself is not dereferenced)
  auto hb_of = [&](const Half& h) {
    HBlk x = hb;
    if (use_hb) x.blocks = static_cast<const T*>(hb.blocks) + (int64_t)h.b0 * hb.bstride;
    return x;
  };
  const int hbm = !use_hb ? 0 : (hb_sc ? HB_MODE_SCATTER : HB_MODE_ROUNDS);   // the off-diagonal kernels' HB template argument
  auto launch_off = [&](const Half& h, int j, int i_first, int nrt) {
    const int Bpad = (h.nb + 7) / 8 * 8;
    const int64_t mo = (int64_t)h.b0 * lstride, po = (int64_t)h.b0 * ntiles * TILE * TILE;
    const T* Hh = use_hb ? nullptr : (const T*)H + (int64_t)h.b0 * hstride;
    auto go = [&](auto mode) {
      constexpr int M = decltype(mode)::value;
      if constexpr (sizeof(T) == 4) {
        hipLaunchKernelGGL(chol_offdiag_f32_kernel<M>, dim3(Bpad * nrt), dim3(256), OFF32_SMEM, h.s, (const float*)Hh, (float*)L + mo,
                           (const float*)panel + po, n, ld, j, ntiles, i_first, nrt, h.nb, pat, hb_of(h));
      } else {
        if constexpr (M != HB_MODE_ROUNDS) {
          if (!tp && !packed && j < f64_half_max) {   // (half tiles, four workgroups per CU, for the short K-loops)
            hipLaunchKernelGGL(chol_offdiag_f64h_kernel<M>, dim3(Bpad * nrt * 2), dim3(256), OFF64_STAGE, h.s, (const double*)Hh,
                               (double*)L + mo, (const double*)panel + po, n, ld, j, ntiles, i_first, nrt, h.nb, pat, hb_of(h));
            return;
          }
          if (!tp && j < f64_wide_max) {   // (dense schedule: column j's K-loops are j tiles long)
            hipLaunchKernelGGL(chol_offdiag_f64w8_kernel<M>, dim3(Bpad * nrt), dim3(512), OFF64_SMEM, h.s, (const double*)Hh,
                               (double*)L + mo, (const double*)panel + po, n, ld, j, ntiles, i_first, nrt, h.nb, pat, hb_of(h));
            return;
          }
        }
        hipLaunchKernelGGL(chol_offdiag_f64_kernel<M>, dim3(Bpad * nrt), dim3(256), OFF64_SMEM, h.s, (const double*)Hh, (double*)L + mo,
                           (const double*)panel + po, n, ld, j, ntiles, i_first, nrt, h.nb, pat, hb_of(h));
      }
    };
    if (hbm == 0) go(std::integral_constant<int, 0>{});
    else if (hbm == HB_MODE_SCATTER) go(std::integral_constant<int, HB_MODE_SCATTER>{});
    else go(std::integral_constant<int, HB_MODE_ROUNDS>{});
  };
  // tiles (i, j) and (i, j + 1) of row tiles [i_first, i_first + nrt) in one workgroup each (chol_offdiag2_f32_kernel)
  auto launch_pair = [&](const Half& h, int j, int i_first, int nrt) {
    if constexpr (sizeof(T) == 4) {
      const int Bpad = (h.nb + 7) / 8 * 8;
      const int64_t mo = (int64_t)h.b0 * lstride, po = (int64_t)h.b0 * ntiles * TILE * TILE;
      const float* Hh = use_hb ? nullptr : (const float*)H + (int64_t)h.b0 * hstride;
      auto go = [&](auto mode) {
        constexpr int M = decltype(mode)::value;
        hipLaunchKernelGGL(chol_offdiag2_f32_kernel<M>, dim3(Bpad * nrt), dim3(256), OFF2_SMEM, h.s, Hh, (float*)L + mo,
                           (const float*)panel + po, n, ld, j, ntiles, i_first, nrt, h.nb, hb_of(h));
      };
      if (hbm == 0) go(std::integral_constant<int, 0>{});
      else if (hbm == HB_MODE_SCATTER) go(std::integral_constant<int, HB_MODE_SCATTER>{});
      else go(std::integral_constant<int, HB_MODE_ROUNDS>{});
    }
  };
  // (nc > 1: the level schedule -- block columns [j, j + nc) in one launch, blockIdx.y the column; `fused` / `smem`: which of the
  //  two diagonal schedules this launch takes, see fused_diag)
  auto launch_diag_n = [&](const Half& h, int j, int nc, bool fused, size_t smem) {
    const int64_t mo = (int64_t)h.b0 * lstride, po = (int64_t)h.b0 * ntiles * TILE * TILE;
    const T* rh = rhs ? (const T*)rhs + (int64_t)h.b0 * ldv : nullptr;
    T* yh = y ? (T*)y + (int64_t)h.b0 * ldv : nullptr;
    const T* dh = damping ? (const T*)damping + h.b0 : nullptr;
    const T* Hh = use_hb ? nullptr : (const T*)H + (int64_t)h.b0 * hstride;
    if (fused) {
      if (use_hb)
        hipLaunchKernelGGL((chol_diag_kernel<T, true>), dim3(h.nb, nc), dim3(256), smem, h.s, Hh, (T*)L + mo, (T*)panel + po,
                           dh, ellipsoidal, (T)eps, info + h.b0, n, ld, j, ntiles, rh, yh, ldv, pat, hb_of(h));
      else
        hipLaunchKernelGGL((chol_diag_kernel<T, false>), dim3(h.nb, nc), dim3(256), smem, h.s, Hh, (T*)L + mo, (T*)panel + po,
                           dh, ellipsoidal, (T)eps, info + h.b0, n, ld, j, ntiles, rh, yh, ldv, pat, hb_of(h));
    } else {
      if (use_hb)
        hipLaunchKernelGGL((chol_syrk_kernel<T, true>), dim3(h.nb, nc), dim3(256), smem, h.s, Hh, (T*)L + mo, dh, ellipsoidal,
                           (T)eps, n, ld, j, rh, yh, ldv, pat, hb_of(h));
      else
        hipLaunchKernelGGL((chol_syrk_kernel<T, false>), dim3(h.nb, nc), dim3(256), smem, h.s, Hh, (T*)L + mo, dh, ellipsoidal,
                           (T)eps, n, ld, j, rh, yh, ldv, pat, hb_of(h));
      const int64_t tile_off = packed ? (int64_t)j * TILE * TILE : (int64_t)j * TILE * ld + (int64_t)j * TILE;
      hipLaunchKernelGGL(chol_potrf_kernel<T>, dim3(h.nb, nc), dim3(64), 0, h.s, (T*)L + mo, (T*)panel + po, info + h.b0, n, lstride,
                         tile_off, packed ? (int64_t)TILE : ld, j, ntiles, rh ? yh : nullptr, ldv, pat.tile_valid);
    }
  };
  auto launch_diag = [&](const Half& h, int j) { launch_diag_n(h, j, 1, fused_diag, dsm); };
  // LEVEL SCHEDULE (thx_chol_factor_levels): the block columns of one elimination-tree level do not depend on each other (a column
  // needs, of the earlier columns, only those in which its own row panel is non-zero: its descendants in the tree) and the host
  // numbered the columns level by level -- so: ONE diagonal launch and ONE off-diagonal launch per level, B x (columns of the
  // level) and B x (entries of the level) workgroups.  A banded ordering's chain of ntiles dependent launch pairs becomes
  // ~log2(ntiles) of them under a nested-dissection ordering (theseus_amd/sparse.py:LevelPattern).
  if (ls) {
    pat.lpt = 1;   // (the level's entries are sorted longest K-list first; consecutive workgroups = the problems of one entry)
    // TWO HALF BATCHES ON TWO STREAMS (THX_LEVEL_SPLIT_MIN=<problems> turns it on; default OFF): a level is SYRK (memory) -> potrf
    // (one wave per tile in its pivot chain: latency) -> off-diagonal tiles (matrix cores); one half's potrf could run underneath the
    // other half's off-diagonal launch (kernel trace at 4096 poses / batch 256: potrf 1.9 ms + SYRK 2.1 ms of a 9.3 ms
    // factorisation with the matrix cores idle).  MEASURED, NOT A WIN (profiles/r6/c_bench_sparse_two_streams.txt: factor 9.00 vs
    // 9.16 ms at batch 256, 2.81 vs 2.76 ms at batch 64; the LM iteration 14.3 vs 13.9 / 5.66 vs 5.61 ms): a level's launches are
    // one to two rounds of workgroups, halving them halves the occupancy of each.  The halves are disjoint problems: no dependence
    // between the streams, same arithmetic.
    static const int level_split_min = [] {
      const char* e = getenv("THX_LEVEL_SPLIT_MIN");
      return e ? atoi(e) : 0;
    }();
    const bool two = level_split_min > 0 && B >= level_split_min && B >= 16;
    Half hv[2] = {{st, 0, B}, {st, 0, 0}};
    if (two) {
      if (!ds.ev_fork) hipEventCreateWithFlags(&ds.ev_fork, hipEventDisableTiming);
      if (!ds.aux[0]) {
        hipStreamCreateWithFlags(&ds.aux[0], hipStreamNonBlocking);
        hipEventCreateWithFlags(&ds.ev_lag[0], hipEventDisableTiming);
        hipEventCreateWithFlags(&ds.ev_join[0], hipEventDisableTiming);
      }
      const int per = min((B / 2 + 7) / 8 * 8, B);
      hv[0] = {st, 0, per};
      hv[1] = {ds.aux[0], per, B - per};
      hipEventRecord(ds.ev_fork, st);
      hipStreamWaitEvent(ds.aux[0], ds.ev_fork, 0);
    }
    // SUBTREE STREAMS (level_stream_host; not together with the half-batch split): levels of group 1 on the second stream, one
    // diagonal phase behind group 0, joined in front of the first trunk level
    const int32_t* lstr = two ? nullptr : ls->level_stream_host;
    static const bool chains_on = [] {
      const char* e = getenv("THX_LEVEL_CHAINS");
      return e ? atoi(e) != 0 : true;
    }();
    if (!chains_on) lstr = nullptr;
    bool forked = false, lag_recorded = false, lag_waited = false;
    if (lstr) {
      if (!ds.ev_fork) hipEventCreateWithFlags(&ds.ev_fork, hipEventDisableTiming);
      if (!ds.aux[0]) {
        hipStreamCreateWithFlags(&ds.aux[0], hipStreamNonBlocking);
        hipEventCreateWithFlags(&ds.ev_lag[0], hipEventDisableTiming);
        hipEventCreateWithFlags(&ds.ev_join[0], hipEventDisableTiming);
      }
      hipEventRecord(ds.ev_fork, st);
      hipStreamWaitEvent(ds.aux[0], ds.ev_fork, 0);
      forked = true;
    }
    for (int l = 0; l < ls->nlevels; ++l) {
      const int j0 = ls->level_col_host[l], nc = ls->level_col_host[l + 1] - j0;
      const int e0 = ls->level_ent_host[l], ne = ls->level_ent_host[l + 1] - e0;
      if (nc <= 0) continue;
      const int yp = rhs ? ls->level_maxk_host[l] * TILE : 0;
      if (lstr) {
        const int code = lstr[l];
        if (forked && (code & 4)) {   // the trunk: everything below it has to be there
          hipEventRecord(ds.ev_join[0], ds.aux[0]);
          hipStreamWaitEvent(st, ds.ev_join[0], 0);
          forked = false;
        }
        const bool second = forked && (code & 3) == 1;
        const Half h{second ? ds.aux[0] : st, 0, B};
        const bool fused = (int64_t)B * nc < split_diag_min;
        if (second && !lag_waited && lag_recorded) {   // group 1 starts one diagonal phase behind group 0
          hipStreamWaitEvent(h.s, ds.ev_lag[0], 0);
          lag_waited = true;
        }
        launch_diag_n(h, j0, nc, fused, fused ? DiagSmem<T>::bytes(yp) : SyrkSmem<T>::bytes(yp));
        if (!second && forked && !lag_recorded) {
          hipEventRecord(ds.ev_lag[0], st);
          lag_recorded = true;
        }
        if (ne > 0) launch_off(h, j0, e0, ne);
        continue;
      }
      for (int k = 0; k < (two ? 2 : 1); ++k) {
        const Half& h = hv[k];
        if (h.nb <= 0) continue;
        // (the schedule of the diagonal phase follows the LAUNCH's workgroup count -- whole batch, so that a problem's arithmetic
        //  does not depend on which half it is in)
        const bool fused = (int64_t)B * nc < split_diag_min;
        if (two && k == 1 && l == 0) hipStreamWaitEvent(h.s, ds.ev_lag[0], 0);   // half 1: one diagonal phase behind half 0
        launch_diag_n(h, j0, nc, fused, fused ? DiagSmem<T>::bytes(yp) : SyrkSmem<T>::bytes(yp));
        if (two && k == 0 && l == 0) hipEventRecord(ds.ev_lag[0], h.s);
        if (ne > 0) launch_off(h, j0, e0, ne);
      }
    }
    if (two || forked) {
      hipEventRecord(ds.ev_join[0], ds.aux[0]);
      hipStreamWaitEvent(st, ds.ev_join[0], 0);
    }
    return check_launch("thx_chol_factor_levels");
  }
  // LOOK-AHEAD for batches that do not fill the chip (one part, i.e. B < THX_CHOL_SPLIT_MIN; THX_CHOL_LOOKAHEAD=0 turns it off).
  // Left-looking: tile (i, j) needs rows i and j of the columns before j.  So the diagonal phase of column j + 1 needs, of column
  // j, only tile (j + 1, j) -- and at batch 256 that phase is B workgroups with one busy wave each (80 us on a 3072-column banded
  // system, 24 times).  Column j's off-diagonal launch is therefore split: HEAD = tile (j + 1, j) on the caller's stream, followed
  // at once by diag(j + 1); REST = the other row tiles on the auxiliary stream, concurrent with both.  Dependencies:
  //   REST(j)  after diag(j)                     (event ev_diag; the earlier HEADs precede diag(j) on the caller's stream)
  //   HEAD(j)  after diag(j) and REST(j - 1)     (row j + 1 of column j - 1 is a REST tile: event ev_rest)
  //   diag(j+1) after HEAD(j) [stream order] -- its other inputs, rows j + 1 of columns < j, were waited for by HEAD(j).
  // Tile-sparse: only if the host put tile (j + 1, j) FIRST in column j's entry list (col_head_host); otherwise the column runs
  // as before (diag(j + 1) waits for all of column j).  Same kernels, same arithmetic: bit-identical results.
  static const bool lookahead_cfg = [] {
    const char* e = getenv("THX_CHOL_LOOKAHEAD");
    return e ? atoi(e) != 0 : true;
  }();
  // Measured (profiles/r4/c_ab_lookahead_small_batch_factor.txt, same box, two rounds): the banded reduced camera system of the
  // bundle-adjustment config (3072 columns, batch 256, 158 of 300 tiles) 6.82 -> 6.60 ms; DENSE frames do not gain (n = 1536:
  // batch 256 3.5 ms either way, batch 512 6.2 -> 6.4 ms; n = 3072 batch 256 21.6 -> 21.8 ms: REST(j) of a dense column is most
  // of the launch, the dispatcher does not run the two queues side by side) -- so: tile-sparse only.
  // DENSE frames between the right-looking schedule's batches and THX_CHOL_LOOKAHEAD_DENSE_MAX_BATCH problems (experiment, default
  // 0 = off): the launches of a block column do not fill the chip there either (batch 64: diag(j) is 64 workgroups, REST(j)
  // 64 (10 - j)).  MEASURED, NOT A WIN (profiles/r6/ai_: batch 64 1.66 -> 1.88 ms, 128 2.18 -> 2.22, 256 3.16 -> 3.27): the whole
  // off-diagonal launch is one round of workgroups, HEAD(j) alone takes as long -- the chain is the serial K-loops.
  static const int dense_la_max = [] {
    const char* e = getenv("THX_CHOL_LOOKAHEAD_DENSE_MAX_BATCH");
    return e ? atoi(e) : 0;
  }();
  const bool dense_la = !tp && !packed && B > rl_max_batch && B <= dense_la_max;
  const bool lookahead = lookahead_cfg && !split && ntiles > 2 && ((tp && tp->col_head_host != nullptr) || dense_la);
  static const bool lpt_cfg = [] {
    const char* e = getenv("THX_CHOL_LPT");
    return e ? atoi(e) != 0 : true;
  }();
  pat.lpt = (lookahead && lpt_cfg && tp) ? 1 : 0;
  if (lookahead) {
    if (!ds.ev_fork) hipEventCreateWithFlags(&ds.ev_fork, hipEventDisableTiming);
    if (!ds.ev_diag) {
      hipEventCreateWithFlags(&ds.ev_diag, hipEventDisableTiming);
      hipEventCreateWithFlags(&ds.ev_rest, hipEventDisableTiming);
    }
    if (!ds.aux[0]) {
      hipStreamCreateWithFlags(&ds.aux[0], hipStreamNonBlocking);
      hipEventCreateWithFlags(&ds.ev_lag[0], hipEventDisableTiming);
      hipEventCreateWithFlags(&ds.ev_join[0], hipEventDisableTiming);
    }
    const Half h0 = halves[0];
    const Half h1 = {ds.aux[0], 0, B};
    hipEventRecord(ds.ev_fork, st);
    hipStreamWaitEvent(h1.s, ds.ev_fork, 0);
    bool rest_pending = false;   // a REST launch whose completion the caller's stream has not waited for yet
    for (int j = 0; j < ntiles; ++j) {
      launch_diag(h0, j);
      const int nrt = tp ? tp->col_count_host[j] : ntiles - 1 - j;
      if (nrt <= 0) continue;
      const bool head = tp ? tp->col_head_host[j] != 0 : true;   // the launch's first tile is (j + 1, j)
      const int n_head = head ? 1 : 0, n_rest = nrt - n_head;
      if (n_rest > 0) {
        hipEventRecord(ds.ev_diag, h0.s);
        hipStreamWaitEvent(h1.s, ds.ev_diag, 0);
      }
      if (rest_pending) {   // HEAD(j) / the next diagonal phase read REST(j - 1)'s tiles
        hipStreamWaitEvent(h0.s, ds.ev_rest, 0);
        rest_pending = false;
      }
      if (n_head) launch_off(h0, j, tp ? 0 : j + 1, 1);
      if (n_rest > 0) {
        launch_off(h1, j, tp ? n_head : j + 1 + n_head, n_rest);
        hipEventRecord(ds.ev_rest, h1.s);
        rest_pending = true;
        if (!head) {   // no look-ahead for this column: diag(j + 1) needs a tile of this launch
          hipStreamWaitEvent(h0.s, ds.ev_rest, 0);
          rest_pending = false;
        }
      }
    }
    if (rest_pending) hipStreamWaitEvent(st, ds.ev_rest, 0);
    return check_launch("thx_chol_factor");
  }
  // RIGHT-LOOKING SCHEDULE for SMALL dense batches (dense L frame without a tile pattern, whole tiles inside the frame, up to
  // thx_chol_schedule.right_looking_max_batch problems).  Left-looking, block column j is two dependent launches whose workgroups
  // walk K-loops of j tiles -- the diagonal one with ONE workgroup per problem: at 8 ... 64 problems the chip is 3 ... 25 % occupied
  // and a factorisation is the sum of those serial K-loops (n = 1536, batch 8: 1.62 ms, 0.04 of the MFMA peak).  Here every tile
  // product is its own workgroup: per block column  chol_diag (the tile factorisation alone) -> chol_offdiag as the substitution
  // alone, B (ntiles - 1 - j) tiles -> chol_offdiag as the trailing update, B m (m + 1) / 2 tiles each receiving ONE product; the
  // working matrix lives in the L frame (first touched by column 0's update, which reads H), the damping of the later diagonal
  // tiles is added once after that update.  Same tile kernels, another summation order: the factor differs from the left-looking
  // one in the last bits (tests: against LAPACK and against the left-looking solution).  The forward substitution runs as its
  // own kernel afterwards.
  {
    if (!tp && !packed && !split && fused_diag && ntiles >= 3 && B <= rl_max_batch && ld >= (int64_t)ntiles * TILE &&
        (!use_hb || !damping || hb.diag_blk)) {
      if (dsm > ds.attr_diag[ti][0]) {   // (the later columns run the dense-frame instance on the L frame whatever H is)
        hipFuncSetAttribute(reinterpret_cast<const void*>(chol_diag_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dsm);
        ds.attr_diag[ti][0] = dsm;
      }
      const Half h{st, 0, B};
      const int Bpad = (B + 7) / 8 * 8;
      // forward substitution riding on the schedule (vectors with 16-byte rows; else its own kernel afterwards, FACTOR_NEEDS_FORWARD): y starts
      // as a copy of g; chol_diag(j) turns block j into y_j in place, the substitution tiles of column j update the blocks below
      const bool fwd_fused = rhs && (ldv % 4) == 0 && (reinterpret_cast<uintptr_t>(y) % 16) == 0;
      if (fwd_fused) {
        hipMemcpy2DAsync(y, (size_t)ldv * sizeof(T), rhs, (size_t)ldv * sizeof(T), (size_t)n * sizeof(T), (size_t)B, hipMemcpyDeviceToDevice, st);
        rhs = y;
      }
      TilePat p0 = pat, p1 = pat;
      p1.rl = 1;
      if (fwd_fused) {
        p0.rl_y = p1.rl_y = y;
        p0.rl_ldv = p1.rl_ldv = ldv;
      }
      const HBlk nohb{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0};
      const T* Lc = (const T*)L;
      const T* yc = fwd_fused ? (const T*)y : nullptr;
      const size_t dsm0 = DiagSmem<T>::bytes(0);
      // one chol_offdiag launch: nrt workgroup slots per problem (row tiles / update tiles), H from the block list, the dense H
      // frame or the L frame
      auto off = [&](bool hbsrc, const T* Hsrc, int jarg, int i_first, int nrt, const TilePat& pp, hipStream_t so = nullptr) {
        if (!so) so = st;
        auto go = [&](auto mode) {
          constexpr int M = decltype(mode)::value;
          if constexpr (sizeof(T) == 4)
            hipLaunchKernelGGL(chol_offdiag_f32_kernel<M>, dim3(Bpad * nrt), dim3(256), OFF32_SMEM, so, (const float*)(M ? nullptr : Hsrc),
                               (float*)L, (const float*)panel, n, ld, jarg, ntiles, i_first, nrt, B, pp, M ? hb : nohb);
          else
            hipLaunchKernelGGL((chol_offdiag_f64_kernel<M, true>), dim3(Bpad * nrt), dim3(256), OFF64_SMEM, so,
                               (const double*)(M ? nullptr : Hsrc), (double*)L, (const double*)panel, n, ld, jarg, ntiles, i_first, nrt, B,
                               pp, M ? hb : nohb);
        };
        if (!hbsrc) go(std::integral_constant<int, 0>{});
        else if (hbm == HB_MODE_SCATTER) go(std::integral_constant<int, HB_MODE_SCATTER>{});
        else go(std::integral_constant<int, HB_MODE_ROUNDS>{});
      };
      auto upd = [&](int jc, bool first) {
        const int m = ntiles - 1 - jc;
        TilePat pu = pat;
        pu.rl = 2 + jc;
        off(first && use_hb, first ? (const T*)H : Lc, jc, 0, m * (m + 1) / 2, pu);
      };
      static const int rl_la_cfg = [] {
        const char* e = getenv("THX_CHOL_RL_LOOKAHEAD");
        return e ? atoi(e) : -1;
      }();
      const int la = rl_la_cfg < 0 ? (sizeof(T) == 8 ? 2 : 1) : (rl_la_cfg > 2 ? 1 : rl_la_cfg);
      // block column 0: the kernels as they are (no earlier columns), reading H
      launch_diag_n(h, 0, 1, true, dsm);   // (with a right-hand side: y_0 = W_00 g_0 -- kept when the forward substitution is fused)
      off(use_hb, (const T*)H, 0, 1, ntiles - 1, p0);
      int jstart = 1;
      if (la == 1) {
        // (mode 1) block column 1 the same way, straight from H: chol_diag(1) and the substitution tiles (i, 1) read their tile of H
        // (block list or dense frame; the diagonal tile with its damping) and take column 0's update through the one-tile K-loop;
        // column 0's update of the tiles right of column 1 -- which moves H into the L frame -- rides with the substitutions.
        // (The plain schedule's update(0) is 66 tiles per problem at 12 block columns: 528 workgroups at batch 8, 16 more than one round.)
        TilePat pd1 = p0;
        pd1.rl = 1;
        pd1.rl_la = 1;
        if (use_hb)
          hipLaunchKernelGGL((chol_diag_kernel<T, true>), dim3(B, 1), dim3(256), fwd_fused ? dsm : dsm0, st, (const T*)nullptr, (T*)L,
                             (T*)panel, (const T*)damping, ellipsoidal, (T)eps, info, n, ld, 1, ntiles, yc, (T*)(fwd_fused ? y : nullptr),
                             ldv, pd1, hb);
        else
          hipLaunchKernelGGL((chol_diag_kernel<T, false>), dim3(B, 1), dim3(256), fwd_fused ? dsm : dsm0, st, (const T*)H, (T*)L,
                             (T*)panel, (const T*)damping, ellipsoidal, (T)eps, info, n, ld, 1, ntiles, yc, (T*)(fwd_fused ? y : nullptr),
                             ldv, pd1, nohb);
        const int nsub = ntiles - 2;
        TilePat pc1 = pd1;
        pc1.rl_nsub = nsub;
        off(use_hb, (const T*)H, 1, 2, nsub + nsub * (nsub + 1) / 2, pc1);
        if (damping && n > 2 * TILE)
          hipLaunchKernelGGL(rl_damp_kernel<T>, dim3((n - 2 * TILE + 255) / 256, B), dim3(256), 0, st, (T*)L, ld, (const T*)H, ld, hb,
                             (const T*)damping, ellipsoidal, (T)eps, 2 * TILE, n);
        jstart = 2;
      } else {
        upd(0, true);
        if (damping)
          hipLaunchKernelGGL(rl_damp_kernel<T>, dim3((n - TILE + 255) / 256, B), dim3(256), 0, st, (T*)L, ld, (const T*)H, ld, hb,
                             (const T*)damping, ellipsoidal, (T)eps, TILE, n);
      }
      // MODE 1, TWO LAUNCHES PER BLOCK COLUMN (TilePat.rl_la / rl_nsub; fp32's default; mode 0: the three launches of the plain schedule).
      // The chain per column was diag -> substitutions -> trailing update, although the next diagonal phase and the next
      // substitutions need only ONE block column of that update.  From block column 2 on every tile of column j takes column
      // j - 1's update ITSELF -- chol_diag(j) and the substitution tiles (i, j) run a K-loop over the one tile of column j - 1, the
      // left-looking kernels' own path -- and the rest of that update (tiles (i, k), j < k <= i: needed from column j + 1 on) rides
      // in the SAME chol_offdiag launch as column j's substitutions, as extra workgroup slots:
      //   diag(j) [own update]  ->  { substitutions (i, j) [own update]  +  update of column j - 1 on the tiles right of column j }
      // No second stream, no events.  Another summation order for the tiles' last update (one K-loop product added before the
      // substitution instead of a read-modify-write before it): to rounding, as the schedule itself.
      // MODE 2, the second stream (fp64's default): only chol_diag(j) takes its own update; update(j - 1) is launched without tile
      // (j, j) (first slot skipped) on the library's second stream and runs BESIDE diag(j); the substitutions of column j wait for
      // both.  Hides the whole update instead of the part the substitutions cover, for two event hops per column.
      // MEASURED (profiles/r6/ao_, n = 1536, factor + forward, plain / mode 1 / mode 2): fp32 batch 8 0.742 / 0.724 / 0.756 ms, 16:
      // 0.856 / 0.828 / 0.855, 32: 1.10 / 1.05 / 1.08; fp64 batch 8 1.53 / 1.42 / 1.34, 16: 1.77 / 1.66 / 1.60, 32: 2.41 / 2.28 / 2.19
      // -- an fp32 update launch is as short as the event hops, an fp64 one twice as long.  THX_CHOL_RL_LOOKAHEAD = 0 | 1 | 2 forces a mode.
      hipStream_t sa = st;
      if (la == 2) {
        if (!ds.ev_diag) {
          hipEventCreateWithFlags(&ds.ev_diag, hipEventDisableTiming);
          hipEventCreateWithFlags(&ds.ev_rest, hipEventDisableTiming);
        }
        if (!ds.aux[0]) {
          hipStreamCreateWithFlags(&ds.aux[0], hipStreamNonBlocking);
          hipEventCreateWithFlags(&ds.ev_lag[0], hipEventDisableTiming);
          hipEventCreateWithFlags(&ds.ev_join[0], hipEventDisableTiming);
        }
        sa = ds.aux[0];
      }
      TilePat pd = p1;
      pd.rl_la = 1;
      bool upd_pending = false;   // (mode 2) an update launch on the second stream that the caller's stream has not waited for yet
      for (int j = jstart; j < ntiles; ++j) {
        hipLaunchKernelGGL((chol_diag_kernel<T, false>), dim3(B, 1), dim3(256), fwd_fused ? dsm : dsm0, st, Lc, (T*)L,
                           (T*)panel, (const T*)nullptr, 0, T(0), info, n, ld, j, ntiles, yc, (T*)(fwd_fused ? y : nullptr), ldv,
                           (la && j >= 2) ? pd : p1, nohb);
        if (j + 1 == ntiles) break;
        if (la == 0) {
          off(false, Lc, j, j + 1, ntiles - 1 - j, p1);
          upd(j, false);
        } else if (la == 1) {
          if (j == 1) {   // (column 1's tiles carry column 0's update already; its own update is folded into column 2's launches)
            off(false, Lc, 1, 2, ntiles - 2, p1);
          } else {
            const int nsub = ntiles - 1 - j;
            TilePat pc = pd;
            pc.rl_nsub = nsub;
            off(false, Lc, j, j + 1, nsub + nsub * (nsub + 1) / 2, pc);
          }
        } else {
          if (upd_pending) {   // the substitutions read the tiles update(j - 1) wrote
            hipStreamWaitEvent(st, ds.ev_rest, 0);
            upd_pending = false;
          }
          off(false, Lc, j, j + 1, ntiles - 1 - j, p1);
          const int m = ntiles - 1 - j, nslots = m * (m + 1) / 2 - 1;   // (slot 0 = tile (j + 1, j + 1): left to diag(j + 1))
          if (nslots > 0) {
            hipEventRecord(ds.ev_diag, st);
            hipStreamWaitEvent(sa, ds.ev_diag, 0);
            TilePat pu = pat;
            pu.rl = 2 + j;
            off(false, Lc, j, 1, nslots, pu, sa);
            hipEventRecord(ds.ev_rest, sa);
            upd_pending = true;
          }
        }
      }
      if (upd_pending) hipStreamWaitEvent(st, ds.ev_rest, 0);
      if (int r = check_launch("thx_chol_factor (right-looking)")) return r;
      return (rhs && !fwd_fused) ? FACTOR_NEEDS_FORWARD : 0;   // (the caller runs the forward substitution as its own kernel)
    }
  }
  // (pairs from 128 problems per call on: the pair schedule's chain per two columns is diag, head tile, diag, pair tiles -- one
  //  more dependent launch than two plain columns -- and below ~128 problems the launches are too small to pay for it: n = 1536,
  //  batch 8 / 16 / 32 / 64: 1.62 / 1.64 / 1.67 / 1.85 ms with pairs, 1.42 / 1.44 / 1.50 / 1.73 ms without; 128: 2.25 / 2.23; 256:
  //  3.25 / 3.30 -- profiles/r6/j_ab_small_batch.txt.  Bit-identical either way.)
  static const int pair_min_batch = [] {
    const char* e = getenv("THX_CHOL_COLPAIR_MIN_BATCH");
    return e ? atoi(e) : 128;
  }();
  const bool colpair = column_pairs != 0 && sizeof(T) == 4 && !tp && !packed && B >= pair_min_batch;
  // STAGGERED PIPELINE (THX_CHOL_LAG_COLS=<columns>, THX_CHOL_PIPE=<parts per stream>; experiment, default off): the batch in
  // 2 x PIPE parts, even parts one after the other on the caller's stream, odd parts on the auxiliary stream which starts LAG
  // block columns behind -- so that one stream's early columns (short K-loops: substitution / store bound, 0.4 - 0.7 of peak per
  // executed flop in fp64) share the CUs with the other stream's late columns (long K-loops: matrix-core bound) instead of with
  // each other.  MEASURED, NOT A WIN (profiles/r6/ad_ab_staggered_streams.txt: fp64 91.7 ms -> 91.8 ... 94.7, fp32 43.7 -> 43.9 ... 45.9
  // over LAG 4 / 6 / 8 x PIPE 1 / 2 / 4): two streams' kernels do not interleave on a CU finely enough.
  static const int lag_cols = [] {
    const char* e = getenv("THX_CHOL_LAG_COLS");
    return e ? atoi(e) : 0;
  }();
  static const int pipe_parts = [] {
    const char* e = getenv("THX_CHOL_PIPE");
    const int v = e ? atoi(e) : 2;
    return v < 1 ? 1 : (v > 8 ? 8 : v);
  }();
  if (split && nparts == 2 && lag_cols > 0 && lag_cols < ntiles && !tp) {
    const int nq = 2 * pipe_parts;
    const int per = min(((B + nq - 1) / nq + 7) / 8 * 8, B);
    for (int q = 0; q < nq; ++q) {
      const int b0 = q * per, nb = min(per, B - b0);
      if (nb <= 0) break;
      const Half h{(q & 1) ? ds.aux[0] : st, b0, nb};
      if (q == 1) hipStreamWaitEvent(h.s, ds.ev_lag[0], 0);
      for (int j = 0; j < ntiles;) {
        const bool pair = colpair && j + 2 < ntiles;
        launch_diag(h, j);
        if (pair) {
          launch_off(h, j, j + 1, 1);
          launch_diag(h, j + 1);
          launch_pair(h, j, j + 2, ntiles - 2 - j);
        } else if (ntiles - 1 - j > 0) {
          launch_off(h, j, j + 1, ntiles - 1 - j);
        }
        j += pair ? 2 : 1;
        if (q == 0 && j >= lag_cols && j - (pair ? 2 : 1) < lag_cols) hipEventRecord(ds.ev_lag[0], h.s);
      }
    }
    hipEventRecord(ds.ev_join[0], ds.aux[0]);
    hipStreamWaitEvent(st, ds.ev_join[0], 0);
    return check_launch("thx_chol_factor");
  }
  for (int j = 0; j < ntiles;) {
    const bool pair = colpair && j + 2 < ntiles;
    for (int k = 0; k < nparts; ++k) {
      const Half& h = halves[k];
      if (h.nb <= 0) continue;
      if (split && k > 0 && j == 0) hipStreamWaitEvent(h.s, ds.ev_lag[k - 1], 0);  // part k: one diagonal phase behind part k-1
      launch_diag(h, j);
      if (split && k + 1 < nparts && j == 0) hipEventRecord(ds.ev_lag[k], h.s);
      if (pair) {
        launch_off(h, j, j + 1, 1);
        launch_diag(h, j + 1);
        launch_pair(h, j, j + 2, ntiles - 2 - j);
        continue;
      }
      const int nrt = tp ? tp->col_count_host[j] : ntiles - 1 - j;   // (tile-sparse: the column's non-zero row tiles)
      if (nrt > 0) launch_off(h, j, tp ? 0 : j + 1, nrt);
    }
    j += pair ? 2 : 1;
  }
  if (split) {
    for (int k = 0; k + 1 < nparts; ++k) {
      hipEventRecord(ds.ev_join[k], ds.aux[k]);
      hipStreamWaitEvent(st, ds.ev_join[k], 0);
    }
  }
  return check_launch("thx_chol_factor");
}

template <typename T>
static int solve_impl(const void* L, int64_t ld, int n, int B, const void* panel, const void* rhs, void* x,
                      int64_t ldv, bool forward, bool backward, hipStream_t st, const thx_tile_pattern* tp = nullptr,
                      const thx_level_schedule* ls = nullptr) {
  const int ntiles = (n + TILE - 1) / TILE;
  const bool list = tp != nullptr;
  const size_t sm = solve_smem<T>(list ? 0 : ntiles * TILE);
  if (sm > LDS_LIMIT) return fail("thx_chol_solve: n too large for the LDS plan (thx_chol_solve_sparse has no limit)");
  {
    std::lock_guard<std::mutex> guard(g_launch_mutex);
    size_t& attr = launch_state().attr_solve[sizeof(T) == 8];
    if (sm > attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(chol_fwd_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)sm);
      hipFuncSetAttribute(reinterpret_cast<const void*>(chol_bwd_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)sm);
      hipFuncSetAttribute(reinterpret_cast<const void*>(chol_fwd_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)sm);
      hipFuncSetAttribute(reinterpret_cast<const void*>(chol_bwd_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)sm);
      attr = sm;
    }
  }
  const bool packed = ld == 0;
  if (packed && (!list || !tp->row_slot || tp->nslots <= 0))
    return fail("thx_chol_solve: a tile-packed factor (ld = 0) needs thx_chol_solve_sparse and a pattern with slot tables");
  RowPat rp{list ? tp->row_ptr : nullptr, list ? tp->row_tile : nullptr, packed ? tp->row_slot : nullptr, packed ? tp->nslots : 0,
            -1, ls ? ls->tile_valid : nullptr};
  const T* src = (const T*)rhs;
  if (ls) {
    // LEVEL SCHEDULE: one launch per elimination-tree level, one workgroup per (problem, block row of the level) -- forward bottom
    // up (a row pulls from its descendants' blocks of y), backward top down (a row pushes into its descendants' blocks of x)
    if (!list) return fail("thx_chol_solve_levels: needs the tile pattern");
    if (forward) {
      for (int l = 0; l < ls->nlevels; ++l) {
        rp.j0 = ls->level_col_host[l];
        const int nc = ls->level_col_host[l + 1] - rp.j0;
        if (nc > 0)
          hipLaunchKernelGGL((chol_fwd_kernel<T, true>), dim3(B, nc), dim3(256), sm, st, (const T*)L, (const T*)panel, src, (T*)x, n,
                             ld, ldv, ntiles, rp);
      }
      src = (const T*)x;
    }
    if (backward) {
      if (src != (const T*)x)
        hipMemcpy2DAsync(x, (size_t)ldv * sizeof(T), src, (size_t)ldv * sizeof(T), (size_t)n * sizeof(T), (size_t)B,
                         hipMemcpyDeviceToDevice, st);
      for (int l = ls->nlevels - 1; l >= 0; --l) {
        rp.j0 = ls->level_col_host[l];
        const int nc = ls->level_col_host[l + 1] - rp.j0;
        if (nc > 0)
          hipLaunchKernelGGL((chol_bwd_kernel<T, true>), dim3(B, nc), dim3(256), sm, st, (const T*)L, (const T*)panel, (const T*)x,
                             (T*)x, n, ld, ldv, ntiles, rp);
      }
    }
    return check_launch("thx_chol_solve_levels");
  }
  if (forward) {
    if (list)
      hipLaunchKernelGGL((chol_fwd_kernel<T, true>), dim3(B), dim3(256), sm, st, (const T*)L, (const T*)panel, src, (T*)x, n,
                         ld, ldv, ntiles, rp);
    else
      hipLaunchKernelGGL((chol_fwd_kernel<T, false>), dim3(B), dim3(256), sm, st, (const T*)L, (const T*)panel, src, (T*)x, n,
                         ld, ldv, ntiles, rp);
    src = (const T*)x;  // the backward pass then runs in place
  }
  // small batches, dense frame: one launch per block row (chol_bwd_rows_kernel; bit-identical to chol_bwd_kernel) -- up to
  // THX_CHOL_BWD_ROWS_MAX_BATCH problems per call (default 32).  A SMALL gain: chol_bwd_kernel's push already runs 256 threads
  // x 8 loads deep (n = 1536, batch 8: 115 us; block rows: 101 us, twelve launches of 5.5 ... 9 us; no difference from 64
  // problems on, profiles/r6/o_ab_bwd_rows.txt) -- it is chol_fwd_kernel's row dots that take 0.46 ms at any batch size, and
  // the right-looking schedule fuses the forward substitution instead.
  static const int bwd_rows_max = [] {
    const char* e = getenv("THX_CHOL_BWD_ROWS_MAX_BATCH");
    return e ? atoi(e) : 32;
  }();
  if (backward && !list && ntiles >= 3 && B <= bwd_rows_max && B <= 65535) {
    if (src != (const T*)x)
      hipMemcpy2DAsync(x, (size_t)ldv * sizeof(T), src, (size_t)ldv * sizeof(T), (size_t)n * sizeof(T), (size_t)B,
                       hipMemcpyDeviceToDevice, st);
    const size_t smr = solve_smem<T>(0);
    {
      std::lock_guard<std::mutex> guard(g_launch_mutex);
      size_t& attr = launch_state().attr_bwd_rows[sizeof(T) == 8];
      if (smr > attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(chol_bwd_rows_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smr);
        attr = smr;
      }
    }
    for (int jb = ntiles; jb >= 1; --jb)
      hipLaunchKernelGGL((chol_bwd_rows_kernel<T>), dim3(jb < ntiles ? jb : 1, B), dim3(256), smr, st, (const T*)L, (const T*)panel,
                         (T*)x, n, ld, ldv, ntiles, jb);
    return check_launch("thx_chol_solve (block rows)");
  }
  if (backward) {
    if (list)
      hipLaunchKernelGGL((chol_bwd_kernel<T, true>), dim3(B), dim3(256), sm, st, (const T*)L, (const T*)panel, src, (T*)x, n,
                         ld, ldv, ntiles, rp);
    else
      hipLaunchKernelGGL((chol_bwd_kernel<T, false>), dim3(B), dim3(256), sm, st, (const T*)L, (const T*)panel, src, (T*)x, n,
                         ld, ldv, ntiles, rp);
  }
  return check_launch("thx_chol_solve");
}

// factor_impl + (right-looking schedule: return code FACTOR_NEEDS_FORWARD) the forward substitution as its own kernel -- outside factor_impl's
// launch lock, which solve_impl takes itself
template <typename T, typename... A>
static int factor_then_forward(const void* H, int64_t ld, int n, int B, const void* damping, int ellipsoidal, double eps, void* L,
                               void* panel, int32_t* info, const void* rhs, void* y, int64_t ldv, hipStream_t st, A... more) {
  const int r = factor_impl<T>(H, ld, n, B, damping, ellipsoidal, eps, L, panel, info, rhs, y, ldv, st, more...);
  if (r != FACTOR_NEEDS_FORWARD) return r;
  return solve_impl<T>(L, ld, n, B, panel, rhs, y, ldv, true, false, st);
}

}  // namespace thx

using namespace thx;

extern "C" {

static int check_factor_args(const void* H, const void* L, const void* panel, const void* info, int n, int B,
                             int64_t ld) {
  if (!H || !L || !panel || !info) return fail("thx_chol_factor: null pointer");
  if (n <= 0 || B <= 0 || (ld != 0 && (ld < n || (ld % 32) != 0)))
    return fail("thx_chol_factor: need n>0, B>0, ld>=n, ld%32==0 (ld = 0: tile-packed factor)");
  return 0;
}

int thx_chol_factor(const void* H, int64_t ld, int32_t n, int32_t B, const void* damping, int ellipsoidal,
                    double damping_eps, void* L, void* Winv, int32_t* info, int dtype, void* stream, const thx_chol_schedule* schedule) {
  if (int r = check_factor_args(H, L, Winv, info, n, B, ld)) return r;
  THX_DISPATCH(dtype,
               return factor_then_forward<float>(H, ld, n, B, damping, ellipsoidal, damping_eps, L, Winv, info, nullptr,
                                         nullptr, 0, as_stream(stream), nullptr, nullptr, nullptr, schedule),
               return factor_then_forward<double>(H, ld, n, B, damping, ellipsoidal, damping_eps, L, Winv, info, nullptr,
                                          nullptr, 0, as_stream(stream), nullptr, nullptr, nullptr, schedule));
  return 0;
}

int thx_chol_factor_forward(const void* H, int64_t ld, int32_t n, int32_t B, const void* damping, int ellipsoidal,
                            double damping_eps, void* L, void* Winv, int32_t* info, const void* rhs, void* y,
                            int64_t ldv, int dtype, void* stream, const thx_chol_schedule* schedule) {
  if (int r = check_factor_args(H, L, Winv, info, n, B, ld)) return r;
  if (!rhs || !y || ldv < n) return fail("thx_chol_factor_forward: rhs / y / ldv");
  if (rhs == y) return fail("thx_chol_factor_forward: y must not alias rhs");
  THX_DISPATCH(dtype,
               return factor_then_forward<float>(H, ld, n, B, damping, ellipsoidal, damping_eps, L, Winv, info, rhs, y, ldv,
                                         as_stream(stream), nullptr, nullptr, nullptr, schedule),
               return factor_then_forward<double>(H, ld, n, B, damping, ellipsoidal, damping_eps, L, Winv, info, rhs, y, ldv,
                                          as_stream(stream), nullptr, nullptr, nullptr, schedule));
  return 0;
}

int thx_chol_factor_sparse(const void* H, int64_t ld, int32_t n, int32_t B, const void* damping, int ellipsoidal,
                           double damping_eps, void* L, void* Winv, int32_t* info, const void* rhs, void* y, int64_t ldv,
                           const thx_tile_pattern* pattern, int dtype, void* stream, const thx_chol_schedule* schedule) {
  if (int r = check_factor_args(H, L, Winv, info, n, B, ld)) return r;
  if (ld == 0) return fail("thx_chol_factor_sparse: H is a dense frame here (ld >= n); the tile-packed factor goes with thx_chol_factor_hblocks");
  if (!pattern || !pattern->col_ptr || !pattern->col_row || !pattern->tile_kptr || !pattern->tile_k || !pattern->diag_kptr ||
      !pattern->diag_k || !pattern->col_count_host)
    return fail("thx_chol_factor_sparse: incomplete tile pattern");
  if ((rhs == nullptr) != (y == nullptr) || (rhs && ldv < n)) return fail("thx_chol_factor_sparse: rhs / y / ldv");
  if (rhs && rhs == y) return fail("thx_chol_factor_sparse: y must not alias rhs");
  THX_DISPATCH(dtype,
               return factor_impl<float>(H, ld, n, B, damping, ellipsoidal, damping_eps, L, Winv, info, rhs, y, ldv,
                                         as_stream(stream), pattern, nullptr, nullptr, schedule),
               return factor_impl<double>(H, ld, n, B, damping, ellipsoidal, damping_eps, L, Winv, info, rhs, y, ldv,
                                          as_stream(stream), pattern, nullptr, nullptr, schedule));
  return 0;
}

static int solve_dispatch(const void* L, int64_t ld, int32_t n, int32_t B, const void* Winv, const void* rhs, void* x,
                          int64_t ldv, bool fwd, bool bwd, int dtype, void* stream, const thx_tile_pattern* tp = nullptr) {
  if (!L || !Winv || !rhs || !x) return fail("thx_chol_solve: null pointer");
  if (n <= 0 || B <= 0 || (ld != 0 && ld < n) || ldv < n) return fail("thx_chol_solve: bad sizes");
  THX_DISPATCH(dtype, return solve_impl<float>(L, ld, n, B, Winv, rhs, x, ldv, fwd, bwd, as_stream(stream), tp),
               return solve_impl<double>(L, ld, n, B, Winv, rhs, x, ldv, fwd, bwd, as_stream(stream), tp));
  return 0;
}

int thx_chol_solve_sparse(const void* L, int64_t ld, int32_t n, int32_t B, const void* Winv, const void* rhs, void* x,
                          int64_t ldv, int backward_only, const thx_tile_pattern* pattern, int dtype, void* stream) {
  if (!pattern || !pattern->row_ptr || !pattern->row_tile) return fail("thx_chol_solve_sparse: incomplete tile pattern");
  if (pattern->ntiles != (n + TILE - 1) / TILE) return fail("thx_chol_solve_sparse: the pattern is not this matrix's");
  return solve_dispatch(L, ld, n, B, Winv, rhs, x, ldv, !backward_only, true, dtype, stream, pattern);
}

int thx_chol_factor_hblocks(const thx_hblock_layout* layout, const void* Hc, int64_t bstride, int32_t n, int32_t B,
                            const void* damping, int ellipsoidal, double damping_eps, void* L, int64_t ld, void* Winv,
                            int32_t* info, const void* rhs, void* y, int64_t ldv, const thx_tile_pattern* pattern, int dtype,
                            void* stream, const thx_chol_schedule* schedule) {
  if (!layout || !layout->tile_ptr || !layout->piece_blk || !layout->piece_rc || !Hc)
    return fail("thx_chol_factor_hblocks: incomplete block layout");
  if (int r = check_factor_args(Hc, L, Winv, info, n, B, ld)) return r;
  if (layout->ntiles != (n + TILE - 1) / TILE || layout->nvars * layout->bd != n || bstride < (int64_t)layout->nblocks * layout->bd * layout->bd)
    return fail("thx_chol_factor_hblocks: the block layout is not this matrix's");
  if (pattern && (!pattern->col_ptr || !pattern->col_row || !pattern->tile_kptr || !pattern->tile_k || !pattern->diag_kptr ||
                  !pattern->diag_k || !pattern->col_count_host))
    return fail("thx_chol_factor_hblocks: incomplete tile pattern");
  if ((rhs == nullptr) != (y == nullptr) || (rhs && ldv < n)) return fail("thx_chol_factor_hblocks: rhs / y / ldv");
  if (rhs && rhs == y) return fail("thx_chol_factor_hblocks: y must not alias rhs");
  const HBlk hb{Hc, bstride, layout->bd, layout->tile_ptr, layout->piece_blk, layout->piece_rc, layout->diag_blk, layout->max_tile_pieces};
  THX_DISPATCH(dtype,
               return factor_then_forward<float>(nullptr, ld, n, B, damping, ellipsoidal, damping_eps, L, Winv, info, rhs, y, ldv,
                                         as_stream(stream), pattern, &hb, nullptr, schedule),
               return factor_then_forward<double>(nullptr, ld, n, B, damping, ellipsoidal, damping_eps, L, Winv, info, rhs, y, ldv,
                                          as_stream(stream), pattern, &hb, nullptr, schedule));
  return 0;
}

static int check_levels(const thx_tile_pattern* pattern, const thx_level_schedule* ls, const char* who) {
  if (!pattern || !pattern->col_ptr || !pattern->col_row || !pattern->tile_kptr || !pattern->tile_k || !pattern->diag_kptr ||
      !pattern->diag_k || !pattern->row_ptr || !pattern->row_tile || !pattern->tile_sa || !pattern->tile_sb || !pattern->diag_s ||
      !pattern->row_slot || pattern->nslots <= 0 || pattern->ntiles <= 0)
    return fail(who, ": incomplete tile pattern (the level schedule works on the tile-packed factor)");
  if (!ls || ls->nlevels <= 0 || !ls->level_col_host || !ls->level_ent_host || !ls->level_maxk_host || !ls->ent_col || !ls->tile_valid)
    return fail(who, ": incomplete level schedule");
  if (ls->level_col_host[0] != 0 || ls->level_col_host[ls->nlevels] != pattern->ntiles || ls->level_ent_host[0] != 0 ||
      ls->level_ent_host[ls->nlevels] != pattern->nslots - pattern->ntiles)
    return fail(who, ": the level schedule does not cover the pattern's block columns / entries");
  for (int l = 0; l < ls->nlevels; ++l)
    if (ls->level_col_host[l + 1] < ls->level_col_host[l] || ls->level_ent_host[l + 1] < ls->level_ent_host[l] ||
        ls->level_col_host[l + 1] - ls->level_col_host[l] > 65535)
      return fail(who, ": level tables must be non-decreasing, at most 65535 block columns per level");
  return 0;
}

int thx_chol_factor_levels(const thx_hblock_layout* layout, const void* Hc, int64_t bstride, int32_t B, const void* damping,
                           int ellipsoidal, double damping_eps, void* L, void* Winv, int32_t* info, const void* rhs, void* y,
                           int64_t ldv, const thx_tile_pattern* pattern, const thx_level_schedule* schedule, int dtype,
                           void* stream, const thx_chol_schedule* chol_schedule) {
  if (!layout || !layout->tile_ptr || !layout->piece_blk || !layout->piece_rc || !Hc || !L || !Winv || !info || B <= 0)
    return fail("thx_chol_factor_levels: null pointer / incomplete block layout / B <= 0");
  if (int r = check_levels(pattern, schedule, "thx_chol_factor_levels")) return r;
  if (layout->ntiles != pattern->ntiles || bstride < (int64_t)layout->nblocks * layout->bd * layout->bd)
    return fail("thx_chol_factor_levels: the block layout is not this pattern's");
  const int n = pattern->ntiles * TILE;   // (the padded order: every tile is whole, tile_valid says how much of it is matrix)
  if ((rhs == nullptr) != (y == nullptr) || (rhs && ldv < n)) return fail("thx_chol_factor_levels: rhs / y are vectors of the PADDED order (ldv >= ntiles * THX_TILE)");
  if (rhs && rhs == y) return fail("thx_chol_factor_levels: y must not alias rhs");
  const HBlk hb{Hc, bstride, layout->bd, layout->tile_ptr, layout->piece_blk, layout->piece_rc, nullptr, layout->max_tile_pieces};
  THX_DISPATCH(dtype,
               return factor_impl<float>(nullptr, 0, n, B, damping, ellipsoidal, damping_eps, L, Winv, info, rhs, y, ldv,
                                         as_stream(stream), pattern, &hb, schedule, chol_schedule),
               return factor_impl<double>(nullptr, 0, n, B, damping, ellipsoidal, damping_eps, L, Winv, info, rhs, y, ldv,
                                          as_stream(stream), pattern, &hb, schedule, chol_schedule));
  return 0;
}

int thx_chol_solve_levels(const void* L, int32_t B, const void* Winv, const void* rhs, void* x, int64_t ldv, int which,
                          const thx_tile_pattern* pattern, const thx_level_schedule* schedule, int dtype, void* stream) {
  if (!L || !Winv || !rhs || !x || B <= 0 || B > 65535) return fail("thx_chol_solve_levels: null pointer / B out of range");
  if (which < 0 || which > 2) return fail("thx_chol_solve_levels: which = 0 (both), 1 (backward only), 2 (forward only)");
  if (int r = check_levels(pattern, schedule, "thx_chol_solve_levels")) return r;
  const int n = pattern->ntiles * TILE;
  if (ldv < n) return fail("thx_chol_solve_levels: rhs / x are vectors of the PADDED order, ldv >= ntiles * THX_TILE");
  THX_DISPATCH(dtype,
               return solve_impl<float>(L, 0, n, B, Winv, rhs, x, ldv, which != 1, which != 2, as_stream(stream), pattern, schedule),
               return solve_impl<double>(L, 0, n, B, Winv, rhs, x, ldv, which != 1, which != 2, as_stream(stream), pattern, schedule));
  return 0;
}

int thx_vec_gather(const void* src, int64_t lds, void* dst, int64_t ldd, const int32_t* idx, int32_t n, int32_t B, int dtype,
                   void* stream) {
  if (!src || !dst || !idx || n <= 0 || B <= 0 || B > 65535 || src == dst) return fail("thx_vec_gather: bad args");
  dim3 grid((n + 255) / 256, B), block(256);
  THX_DISPATCH(dtype,
               hipLaunchKernelGGL(vec_gather_kernel<float>, grid, block, 0, as_stream(stream), (const float*)src, lds, (float*)dst,
                                  ldd, idx, n),
               hipLaunchKernelGGL(vec_gather_kernel<double>, grid, block, 0, as_stream(stream), (const double*)src, lds,
                                  (double*)dst, ldd, idx, n));
  return check_launch("thx_vec_gather");
}

int thx_chol_solve(const void* L, int64_t ld, int32_t n, int32_t B, const void* Winv, const void* rhs, void* x,
                   int64_t ldv, int dtype, void* stream) {
  return solve_dispatch(L, ld, n, B, Winv, rhs, x, ldv, true, true, dtype, stream);
}

int thx_chol_solve_backward(const void* L, int64_t ld, int32_t n, int32_t B, const void* Winv, const void* y, void* x,
                            int64_t ldv, int dtype, void* stream) {
  return solve_dispatch(L, ld, n, B, Winv, y, x, ldv, false, true, dtype, stream);
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
