// The factorisation's fp32 K-loop rebuilt piece by piece, to see which piece costs what against the bare MFMA rate
// (tools/microbench/mfma_sustained.hip).  One workgroup = 4 waves, each wave a 32 x 128 slab of a 128 x 128 product, k-chunks
// of 32 staged in LDS with row stride 36 words (the layout of Engine<float>::chunk in theseus_amd/csrc/chol_kernels.hip).
//   mode 0: fragment reads + MFMAs, the four MFMAs of one accumulator consecutive (dependent), as the kernel writes them
//   mode 1: the same with the four accumulators interleaved (consecutive MFMAs independent)
//   mode 2: mode 0 + the K-loop's two barriers per chunk
//   mode 3: mode 2 + the register -> LDS staging stores (8 ds_write_b128 per thread and chunk)
//   mode 4: mode 3 + the operand stream from HBM (2 x 128 rows x 32 columns per chunk through buffer-less global loads,
//           one chunk ahead), i.e. the whole K-loop: row-major matrix, leading dimension 1536 (a chunk = 256 pieces of 128 B)
//   mode 9: mode 4 with the fragment reads software-pipelined one k-step ahead (the compiler requests them 4 MFMAs ahead)
//   mode 5: mode 4 with the leading dimension padded to 1536 + 32 words (HBM channel camping?)
//   mode 6: mode 4 with every 128 x 32 chunk CONTIGUOUS in memory (16 KB, tile-packed operand layout)
// each at 2 workgroups per CU (512 workgroups) and at 1 per CU (256 workgroups, LDS padded to force it).
// build: hipcc --offload-arch=gfx950 -O3 -o kloop_pieces tools/microbench/kloop_pieces.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDT = 36, LD = 1536;

template <int MODE, int PAD>
__global__ void __launch_bounds__(256, 2) k(const float* __restrict__ mats, float* out, int chunks) {
  __shared__ float sA[128 * LDT], sB[128 * LDT];
  __shared__ float pad[PAD ? 12 * 1024 : 1];   // 1 workgroup per CU: 36 + 48 KB > 80 KB
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rl = lane & 31, g = lane >> 5;
  unsigned s = 1234567u * (tid + 1) + blockIdx.x * 7919u;
  for (int i = tid; i < 128 * LDT; i += 256) {
    s = s * 1664525u + 1013904223u;
    sA[i] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
    s = s * 1664525u + 1013904223u;
    sB[i] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
  }
  if (PAD) pad[tid] = 0.f;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const float* sBw = sB + 32 * wave * LDT;
  const int lrow = tid >> 3, lc = tid & 7;
  const float* mat = mats + (size_t)(blockIdx.x % 512) * (LD + 32) * LD;
  float4 ra[4], rb[4];
  for (int u = 0; u < 4; ++u) ra[u] = rb[u] = make_float4(0.1f, 0.2f, 0.3f, 0.4f);
  auto gload = [&](int kc) __attribute__((always_inline)) {
    const int rt = 1 + (kc / 44) % 11, kq = kc % 44, k0 = kq * 32;   // row tile rt against row tile 0, columns k0 .. k0+31
    constexpr int ld = MODE == 5 ? LD + 32 : LD;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 6) {   // chunk (row tile, kq) = 4096 contiguous words
        ra[u] = *reinterpret_cast<const float4*>(mat + ((size_t)rt * 48 + kq) * 4096 + 4 * (tid + 256 * u));
        rb[u] = *reinterpret_cast<const float4*>(mat + (size_t)kq * 4096 + 4 * (tid + 256 * u));
      } else {
        ra[u] = *reinterpret_cast<const float4*>(mat + (size_t)(128 * rt + lrow + 32 * u) * ld + k0 + 4 * lc);
        rb[u] = *reinterpret_cast<const float4*>(mat + (size_t)(lrow + 32 * u) * ld + k0 + 4 * lc);
      }
    }
  };
  if (MODE >= 4) gload(0);
  for (int kc = 0; kc < chunks; ++kc) {
    asm volatile("" ::: "memory");   // the fragment reads stay inside the loop
    if (MODE >= 2) __syncthreads();
    if (MODE >= 3) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        *reinterpret_cast<float4*>(sA + (lrow + 32 * u) * LDT + 4 * lc) = ra[u];
        *reinterpret_cast<float4*>(sB + (lrow + 32 * u) * LDT + 4 * lc) = rb[u];
      }
    }
    if (MODE >= 2) __syncthreads();
    if (MODE >= 4) {
      gload(kc + 1);
      __builtin_amdgcn_sched_barrier(0);   // the loads are issued BEFORE the MFMAs (the compiler sinks them below otherwise)
    }
    if (MODE == 9) {   // fragments of k-step ks+1 are requested before the 16 MFMAs of k-step ks
      float4 fbq[2], faq[2][4];
      auto frag = [&](int ks, float4& fb, float4 (&fa)[4]) __attribute__((always_inline)) {
        fb = *reinterpret_cast<const float4*>(sBw + rl * LDT + 8 * ks + 4 * g);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) fa[cb] = *reinterpret_cast<const float4*>(sA + (32 * cb + rl) * LDT + 8 * ks + 4 * g);
      };
      frag(0, fbq[0], faq[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) frag(ks + 1, fbq[(ks + 1) & 1], faq[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const float4 fb = fbq[ks & 1];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          const float4 fa = faq[ks & 1][cb];
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc[cb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      continue;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float4 fb = *reinterpret_cast<const float4*>(sBw + rl * LDT + 8 * ks + 4 * g);
      if (MODE == 1) {
        float4 fa[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) fa[cb] = *reinterpret_cast<const float4*>(sA + (32 * cb + rl) * LDT + 8 * ks + 4 * g);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb].x, fb.x, acc[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb].y, fb.y, acc[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb].z, fb.z, acc[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb].w, fb.w, acc[cb], 0, 0, 0);
      } else {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          const float4 fa = *reinterpret_cast<const float4*>(sA + (32 * cb + rl) * LDT + 8 * ks + 4 * g);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc[cb], 0, 0, 0);
        }
      }
    }
  }
  float t = PAD ? pad[tid] : 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) t += acc[i][j];
  out[blockIdx.x * 256 + tid] = t + ra[0].x + rb[3].w;
}

// mode 7: the operand stream lands in LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write), two 32 KB
// chunk buffers, rows unpadded with the 16-byte quads XOR-swizzled by (row >> 1) & 7 on both sides, ONE barrier per chunk.
template <int PIN>
__global__ void __launch_bounds__(256, 2) k_glds(const float* __restrict__ mats, float* out, int chunks) {
  const float* mat = mats + (size_t)(blockIdx.x % 512) * (LD + 32) * LD;
  __shared__ float4 buf0[2048], buf1[2048];   // each: A 128 rows x 8 quads, B 128 rows x 8 quads
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rl = lane & 31, g = lane >> 5;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  auto issue = [&](float4* buf, int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = 32 * u + 8 * wave + (lane >> 3), q = (lane & 7) ^ ((r >> 1) & 7);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mat + (size_t)(128 * (1 + (kc / 44) % 11) + r) * LD + (kc % 44) * 32 + 4 * q),
                                       (__attribute__((address_space(3))) void*)(buf + (32 * u + 8 * wave) * 8), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mat + (size_t)r * LD + (kc % 44) * 32 + 4 * q),
                                       (__attribute__((address_space(3))) void*)(buf + 1024 + (32 * u + 8 * wave) * 8), 16, 0, 0);
    }
  };
  auto compute = [&](const float4* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int rb = 32 * wave + rl;
      const float4 fb = buf[1024 + rb * 8 + ((2 * ks + g) ^ ((rb >> 1) & 7))];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const int ra = 32 * cb + rl;
        const float4 fa = buf[ra * 8 + ((2 * ks + g) ^ ((ra >> 1) & 7))];
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc[cb], 0, 0, 0);
      }
    }
  };
  issue(buf0, 0);
  __syncthreads();
  for (int kc = 0; kc < chunks; kc += 2) {
    issue(buf1, kc + 1);
    if (PIN) __builtin_amdgcn_sched_barrier(0);
    compute(buf0);
    __syncthreads();
    issue(buf0, kc + 2);
    if (PIN) __builtin_amdgcn_sched_barrier(0);
    compute(buf1);
    __syncthreads();
  }
  float t = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) t += acc[i][j];
  out[blockIdx.x * 256 + tid] = t;
}

template <int MODE, int PAD>
void run(const char* name, const float* mats, float* out) {
  const int wgs = PAD ? 256 : 512, chunks = 8000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  if (MODE == 7 || MODE == 8) k_glds<MODE - 7><<<wgs, 256>>>(mats, out, 64); else k<(MODE == 7 || MODE == 8) ? 0 : MODE, PAD><<<wgs, 256>>>(mats, out, 64);
  hipDeviceSynchronize();
  float best = 1e30f, sum = 0;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    if (MODE == 7 || MODE == 8) k_glds<MODE - 7><<<wgs, 256>>>(mats, out, chunks); else k<(MODE == 7 || MODE == 8) ? 0 : MODE, PAD><<<wgs, 256>>>(mats, out, chunks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double flops = (double)wgs * 4 * chunks * 64 * 4096.0;
  printf("%-62s %d WG/CU: %7.2f ms  %6.1f TFLOP/s  (%.3f of 157.3; a chunk = %5.0f cycles at 2.35 GHz against 4096 x %d of MFMA issue)\n",
         name, PAD ? 1 : 2, sum / 4, flops / (sum / 4) / 1e9, flops / (sum / 4) / 1e9 / 157.3, sum / 4 * 1e-3 * 2.35e9 / chunks, PAD ? 1 : 2);
}

int main() {
  float *mats, *out;
  const size_t n = (size_t)512 * (LD + 32) * LD;
  hipMalloc(&mats, n * 4);
  hipMemset(mats, 0x3c, n * 4);   // 0x3c3c3c3c = 0.0115 as fp32
  hipMalloc(&out, 512 * 256 * 4);
  run<0, 0>("0 reads + MFMAs (4 dependent MFMAs in a row)", mats, out);
  run<1, 0>("1 reads + MFMAs (accumulators interleaved)", mats, out);
  run<2, 0>("2 = 0 + two barriers per chunk", mats, out);
  run<3, 0>("3 = 2 + register -> LDS staging stores", mats, out);
  run<4, 0>("4 = 3 + operand stream from HBM (the whole K-loop)", mats, out);
  run<5, 0>("5 = 4, leading dimension 1536 + 32", mats, out);
  run<6, 0>("6 = 4, chunks contiguous (tile-packed operands)", mats, out);
  run<9, 0>("9 = 4 with the fragment reads one k-step (16 MFMAs) ahead", mats, out);
  run<7, 0>("7 = direct-to-LDS operand stream, double buffered", mats, out);
  run<8, 0>("8 = 7 with the loads pinned in front of the MFMAs", mats, out);
  run<0, 1>("0 reads + MFMAs (4 dependent MFMAs in a row)", mats, out);
  run<1, 1>("1 reads + MFMAs (accumulators interleaved)", mats, out);
  run<2, 1>("2 = 0 + two barriers per chunk", mats, out);
  run<3, 1>("3 = 2 + register -> LDS staging stores", mats, out);
  run<4, 1>("4 = 3 + operand stream from HBM (the whole K-loop)", mats, out);
  run<5, 1>("5 = 4, leading dimension 1536 + 32", mats, out);
  run<6, 1>("6 = 4, chunks contiguous (tile-packed operands)", mats, out);
  return 0;
}
