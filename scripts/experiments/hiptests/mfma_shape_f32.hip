// micro-benchmark (round 3): does the fp32 MFMA SHAPE change the clock the chip holds under load?  MI355X_MICROARCH.md
// ('DVFS give-back' item 7) reports 1.15x for the 16x16x32 bf16 shape over 32x32x16 at equal cycles per flop.  Here:
// v_mfma_f32_32x32x2_f32 (4096 flop / 64 cycles) vs v_mfma_f32_16x16x4_f32 (2048 flop / 32 cycles), 64 accumulator
// registers per wave either way, operands in registers, RANDOM data, 256 x 4 waves (one per SIMD) and 256 x 8 waves.
// hipcc --offload-arch=gfx950 -O3 -o mfma_shape_f32 mfma_shape_f32.hip && ./mfma_shape_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NW>
__global__ __launch_bounds__(64 * NW) void shape_kernel(const float *__restrict__ rnd, float *__restrict__ out, unsigned long long *clk, int reps) {
  const int tid = threadIdx.x, gid = blockIdx.x * blockDim.x + tid;
  float a[8], b[8];
  for (int u = 0; u < 8; ++u) { a[u] = rnd[(gid * 16 + u) & 0xfffff]; b[u] = rnd[(gid * 16 + 8 + u) & 0xfffff]; }
  unsigned long long c0 = 0, r0 = 0;
  if (SHAPE == 32) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + j) & 7], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 12345.678f) out[gid] = s;
  } else {
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int u = 0; u < 4; ++u)           // 4 x 16 = 64 MFMAs of 2048 flop = the 32 MFMAs of 4096 flop above
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + j) & 7], b[(2 * u + j) & 7], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) for (int e = 0; e < 4; ++e) s += acc[j][e];
    if (s == 12345.678f) out[gid] = s;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((tid & 63) == 0) { clk[2 * (gid >> 6)] = c1 - c0; clk[2 * (gid >> 6) + 1] = r1 - r0; }
}

template <int SHAPE, int NW>
void run(const char *name, const float *rnd, float *out, unsigned long long *clk) {
  const int wgs = 256, reps = 60000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int warm = 0; warm < 3; ++warm) hipLaunchKernelGGL((shape_kernel<SHAPE, NW>), dim3(wgs), dim3(64 * NW), 0, 0, rnd, out, clk, reps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 4; ++it) hipLaunchKernelGGL((shape_kernel<SHAPE, NW>), dim3(wgs), dim3(64 * NW), 0, 0, rnd, out, clk, reps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * wgs * NW);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double ghz = 0; for (int w = 0; w < wgs * NW; ++w) ghz += (double)h[2 * w] / (double)h[2 * w + 1] * 0.1;
  ghz /= wgs * NW;
  const double flop = 4.0 * wgs * NW * (double)reps * 32 * 4096.0;
  printf("%-28s %8.2f ms  %7.1f TFLOP/s  in-kernel clock %.3f GHz  -> %.1f flop/clk/SIMD\n", name, ms, flop / (ms * 1e-3) / 1e12, ghz,
         flop / (ms * 1e-3) / (ghz * 1e9) / 1024.0);
}

int main() {
  float *rnd, *out; unsigned long long *clk;
  std::vector<float> h(1 << 20);
  srand(7); for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMalloc(&rnd, h.size() * 4); hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&out, 1 << 22); hipMalloc(&clk, 1 << 20);
  for (int pass = 0; pass < 2; ++pass) {
    run<32, 4>("32x32x2  1 wave/SIMD", rnd, out, clk);
    run<16, 4>("16x16x4  1 wave/SIMD", rnd, out, clk);
    run<32, 8>("32x32x2  2 waves/SIMD", rnd, out, clk);
    run<16, 8>("16x16x4  2 waves/SIMD", rnd, out, clk);
  }
  return 0;
}
