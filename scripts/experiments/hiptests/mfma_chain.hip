// micro-benchmark: throughput of v_mfma_f32_32x32x2_f32 issued as DEPENDENT chains (one accumulator per wave)
// vs independent accumulators, at 1..6 waves per SIMD.  Explains what mlp_chain_reg_kernel can reach.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void chain(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < NACC; ++n)
    for (int e = 0; e < 16; ++e) s += acc[n][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd) {
  float *d;
  const int blocks = 256 * waves_per_simd, threads = 256;   // 4 waves per block -> one per SIMD
  hipMalloc(&d, sizeof(float) * blocks * threads);
  const int iters = 2000 / NACC;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(chain<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f, 1e-6f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f, 1e-6f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 16 * NACC * 4096.0;
  printf("accumulators per wave %d, waves/SIMD %d: %.1f TF\n", NACC, waves_per_simd, flops / ms / 1e9);
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 4; ++w) run<1>(w);
  for (int w = 1; w <= 4; ++w) run<2>(w);
  run<4>(1); run<4>(2);
  return 0;
}
