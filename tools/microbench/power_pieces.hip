// What the pieces of the fp32 K-loop cost in SOCKET POWER and SHADER CLOCK (the MI355X runs under a 1400 W cap; rocm-smi
// reports 1356 W / 2.2 GHz under thx_chol_factor -- profiles/r5/q_): each mode runs back to back for a given time while
// tools/power_model.py samples librocm_smi64; every workgroup also measures its own shader clock (s_memtime cycles against
// the 100 MHz wall clock), so the clock does not depend on the SMI's sampling.
//   mode 0: MFMAs only, pseudo-random operands in registers              (v_mfma_f32_32x32x2_f32, 4 accumulators per wave)
//   mode 1: + the fragment reads from LDS at the K-loop's rate           (5 ds_read_b128 per 16 MFMAs)
//   mode 2: + two barriers and the register -> LDS staging stores per k-chunk
//   mode 3: + the operand stream from HBM, one chunk ahead               (= the whole K-loop of chol_offdiag_f32)
//   mode 4: the operand stream alone (loads + staging stores + barriers, NO MFMAs)
//   mode 5: mode 3 with operands that are all 0.0115 (low toggle rate)
// 2 workgroups of 4 waves per CU (512 workgroups).  build: hipcc --offload-arch=gfx950 -O3 -o power_pieces power_pieces.hip
// usage: power_pieces <mode> <seconds>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDT = 36, LD = 1536;

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(const float* __restrict__ mats, float* out, unsigned long long* clk, int chunks) {
  __shared__ float sA[128 * LDT], sB[128 * LDT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rl = lane & 31, g = lane >> 5;
  unsigned s = 1234567u * (tid + 1) + blockIdx.x * 7919u;
  for (int i = tid; i < 128 * LDT; i += 256) {
    s = s * 1664525u + 1013904223u;
    sA[i] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
    s = s * 1664525u + 1013904223u;
    sB[i] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
  }
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const float* sBw = sB + 32 * wave * LDT;
  const int lrow = tid >> 3, lc = tid & 7;
  const float* mat = mats + (size_t)(blockIdx.x % 512) * LD * LD;
  float4 ra[4], rb[4];
  for (int u = 0; u < 4; ++u) ra[u] = rb[u] = make_float4(sA[tid], sB[tid], sA[tid + 256], sB[tid + 256]);
  auto gload = [&](int kc) __attribute__((always_inline)) {
    const int rt = 1 + (kc / 44) % 11, k0 = (kc % 44) * 32;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ra[u] = *reinterpret_cast<const float4*>(mat + (size_t)(128 * rt + lrow + 32 * u) * LD + k0 + 4 * lc);
      rb[u] = *reinterpret_cast<const float4*>(mat + (size_t)(lrow + 32 * u) * LD + k0 + 4 * lc);
    }
  };
  if (MODE >= 3) gload(0);
  float4 fx = ra[0], fy = rb[0];
  for (int kc = 0; kc < chunks; ++kc) {
    asm volatile("" ::: "memory");
    if (MODE >= 2) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        *reinterpret_cast<float4*>(sA + (lrow + 32 * u) * LDT + 4 * lc) = ra[u];
        *reinterpret_cast<float4*>(sB + (lrow + 32 * u) * LDT + 4 * lc) = rb[u];
      }
      __syncthreads();
    }
    if (MODE >= 3) {
      gload(kc + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 4) continue;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float4 fb = fy;
      if (MODE >= 1) fb = *reinterpret_cast<const float4*>(sBw + rl * LDT + 8 * ks + 4 * g);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        float4 fa = fx;
        if (MODE >= 1) fa = *reinterpret_cast<const float4*>(sA + (32 * cb + rl) * LDT + 8 * ks + 4 * g);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc[cb], 0, 0, 0);
      }
    }
  }
  float t = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) t += acc[i][j];
  out[blockIdx.x * 256 + tid] = t + ra[0].x + rb[3].w;
  if (tid == 0) {
    clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
    clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
  }
}

__global__ void fill(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned s = (unsigned)i * 2654435761u + 12345u;
    s = s * 1664525u + 1013904223u;
    s ^= s >> 15;
    p[i] = (s & 0x807fffffu) | (0x3f000000u + ((s >> 7) & 0x00800000u));
  }
}

template <int MODE>
void run(const float* mats, float* out, unsigned long long* clk, double seconds) {
  const int wgs = 512, chunks = MODE == 4 ? 4000 : 8000;
  k<MODE><<<wgs, 256>>>(mats, out, clk, 64);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const auto t0 = std::chrono::steady_clock::now();
  double ms_sum = 0, cyc = 0, wall = 0;
  int launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) k<MODE><<<wgs, 256>>>(mats, out, clk, chunks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms_sum += ms;
    launches += 4;
    unsigned long long h[2 * 512];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    for (int w = 0; w < wgs; ++w) { cyc += (double)h[2 * w]; wall += (double)h[2 * w + 1]; }
  }
  const double ms = ms_sum / launches;
  const double flops = MODE == 4 ? 0.0 : (double)wgs * 4 * chunks * 64 * 4096.0;
  const double bytes = MODE >= 3 ? (double)wgs * chunks * 32768.0 : 0.0;   // operand stream requested per launch (L2 + HBM)
  const double ghz = cyc / (wall * 10.0);   // cycles per 10 ns tick
  printf("mode %d: %d launches, %.2f ms each, %.1f TFLOP/s (%.3f of 157.3), operand stream %.2f TB/s, shader clock %.3f GHz, "
         "%.0f cycles per chunk (MFMA issue: 8192 per chunk pair of a CU)\n",
         MODE, launches, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3, bytes / ms / 1e9, ghz, ms * 1e-3 * ghz * 1e9 / chunks);
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 3;
  const double seconds = argc > 2 ? atof(argv[2]) : 2.0;
  float *mats, *out;
  unsigned long long* clk;
  const size_t n = (size_t)512 * LD * LD;
  hipMalloc(&mats, n * 4);
  hipMalloc(&out, 512 * 256 * 4);
  hipMalloc(&clk, 2 * 512 * 8);
  if (mode == 5) {
    hipMemset(mats, 0x3c, n * 4);
  } else {   // pseudo-random words with a sane exponent: |x| in [0.5, 2)
    fill<<<4096, 256>>>((unsigned*)mats, n);
    hipDeviceSynchronize();
  }
  switch (mode) {
    case 0: run<0>(mats, out, clk, seconds); break;
    case 1: run<1>(mats, out, clk, seconds); break;
    case 2: run<2>(mats, out, clk, seconds); break;
    case 3: case 5: run<3>(mats, out, clk, seconds); break;
    case 4: run<4>(mats, out, clk, seconds); break;
  }
  return 0;
}
