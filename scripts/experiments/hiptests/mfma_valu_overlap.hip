// micro-benchmark: does VALU work of one wave overlap with MFMA work of another wave on the same SIMD?
// Each wave alternates: 16 dependent v_mfma_f32_32x32x2_f32 (1024 cycles of matrix pipe), then NV VALU ops
// that depend on nothing matrix-related.  Reported: cycles per phase pair and per-SIMD matrix-pipe utilisation.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0) {
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a0 + i + threadIdx.x;
  const float a = a0, b = 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < NV / 8; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
  }
  float s = 0;
  for (int e = 0; e < 16; ++e) s += acc[e];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
void run(int waves_per_simd) {
  float *d;
  const int blocks = 256 * waves_per_simd, iters = 400;
  hipMalloc(&d, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_cycles = (double)waves_per_simd * iters * 16 * 64;    // matrix-pipe demand per SIMD
  const double total_cycles = ms * 1e-3 * 2.4e9;
  printf("VALU ops per phase %3d, waves/SIMD %d: %.3f ms, matrix pipe busy %.0f %% (at 2.4 GHz)\n", NV, waves_per_simd, ms,
         100.0 * mfma_cycles / total_cycles);
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 4; ++w) run<0>(w);
  for (int w = 1; w <= 4; ++w) run<128>(w);
  for (int w = 1; w <= 4; ++w) run<256>(w);
  return 0;
}
