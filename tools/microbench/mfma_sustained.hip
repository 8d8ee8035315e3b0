// Sustained fp32 MFMA rate of the whole chip under operand data like the factorisation's: is 157.3 TFLOP/s (256 CUs x 256
// flops/cycle x 2.4 GHz) reachable for tens of milliseconds, or does the part settle lower?  v_mfma_f32_32x32x2_f32, 2
// workgroups of 4 waves per CU, 4 independent accumulators per wave (the K-loop's shape), operands constant (the
// round-1 microbenchmark: 150-156 TFLOP/s in 3.5 ms bursts), pseudo-random per lane, or pseudo-random and changing.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_sustained tools/microbench/mfma_sustained.hip ; run: ./mfma_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>  // 0 constant operands, 1 pseudo-random operands fixed per lane, 2 pseudo-random and changing
__global__ void __launch_bounds__(256, 2) k(float* out, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float x[8], y[8];
  unsigned s = 1234567u * (threadIdx.x + 1) + blockIdx.x * 7919u;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    x[i] = MODE ? (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f : a;
    s = s * 1664525u + 1013904223u;
    y[i] = MODE ? (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f : b;
  }
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // static register indices throughout
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[(r + i) & 7], y[(r + 2 * i) & 7], acc[i], 0, 0, 0);
      if (MODE == 2) {  // keep the operands moving: 4 VALU ops per 32 MFMAs
        x[u] = x[u] * 0.999f + 0.001f * y[(u + 3) & 7];
        y[(u + 5) & 7] = y[(u + 5) & 7] * 0.999f - 0.001f * x[(u + 1) & 7];
      }
    }
  }
  float t = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) t += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

// The K-loop's operand traffic as well: every 4 MFMAs (a 2x2 block of 32x32 accumulators, one k-step) take 2 A and 2 B
// values per lane from LDS (pseudo-random contents), i.e. one ds_read_b128 per 4 MFMAs.
__global__ void __launch_bounds__(256, 2) k_lds(float* out, int iters) {
  __shared__ float4 tile[2048];  // 32 KB
  unsigned s = 1234567u * (threadIdx.x + 1) + blockIdx.x * 7919u;
  for (int i = threadIdx.x; i < 2048; i += 256) {
    float v[4];
    for (int j = 0; j < 4; ++j) {
      s = s * 1664525u + 1013904223u;
      v[j] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    tile[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  int at = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float4 v = tile[(at + r * 256) & 2047];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, v.z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, v.w, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, v.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, v.w, acc[3], 0, 0, 0);
    }
    at = (at + 64) & 2047;
  }
  float t = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) t += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MODE>
void run(const char* name, int iters, int reps) {
  const int wgs = 512;
  float* out;
  hipMalloc(&out, (size_t)wgs * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  if (MODE == 3) k_lds<<<wgs, 256>>>(out, 10); else k<MODE == 3 ? 0 : MODE><<<wgs, 256>>>(out, 16, 1.f, 2.f);
  hipDeviceSynchronize();
  printf("%s: %d launches of %d x 32 MFMAs per wave, back to back\n", name, reps, iters);
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    if (MODE == 3) k_lds<<<wgs, 256>>>(out, iters); else k<MODE == 3 ? 0 : MODE><<<wgs, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * 4 * iters * 32 * 4096.0;
    printf("   launch %2d: %7.2f ms  %6.1f TFLOP/s\n", r, ms, flops / ms / 1e9);
  }
  hipFree(out);
}

int main() {
  run<0>("constant operands", 4000, 3);   // ~7 ms bursts
  run<0>("constant operands", 40000, 6);  // ~70 ms each
  run<1>("random operands, fixed", 4000, 3);
  run<1>("random operands, fixed", 40000, 6);
  run<2>("random operands, changing", 4000, 3);
  run<2>("random operands, changing", 40000, 12);  // ~1 s in total
  run<3>("random operands from LDS (1 ds_read_b128 per 4 MFMAs)", 4000, 3);
  run<3>("random operands from LDS (1 ds_read_b128 per 4 MFMAs)", 40000, 12);
  return 0;
}
