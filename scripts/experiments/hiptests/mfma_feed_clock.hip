// micro-benchmark (round 3): which operand path costs the head group's K loop its clock?  The bare fp32 MFMA holds 2.38 GHz on
// random data (mfma_shape_f32.hip), mlp_group_kernel<256,512,1024> runs at 2.16 GHz.  Same loop structure as the kernel's third
// layer (8 waves per workgroup, 4 accumulator tiles per wave, blocks of 8 k-steps x 4 tiles = 32 MFMAs), RANDOM data:
//   mode 0 registers only | 1 + A fragments from LDS | 2 + B fragments streamed from L2 (dwordx4, two blocks ahead) | 3 both
//   mode 7 = both, with TWO 32-row blocks per workgroup sharing every B fragment (M = 64: half the L2 bytes per flop;
//            8 accumulator tiles per wave, 16 A fragments per block)
// prints wall time, TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
// hipcc --offload-arch=gfx950 -O3 -o mfma_feed_clock mfma_feed_clock.hip && ./mfma_feed_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int RB>
__global__ __launch_bounds__(512) void feed_kernel(const float *__restrict__ w, const float *__restrict__ xin, float *__restrict__ out,
                                                  unsigned long long *clk, int nblk, int reps) {
  extern __shared__ float X[];                     // RB x 32 rows x 513
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  for (int i = tid; i < RB * 32 * 513; i += 512) X[i] = xin[(i + 7 * blockIdx.x) & 0xfffff];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, 0xffffffff, 0x00020000);
  const uint32_t voff = (uint32_t)(kh * 1024 + wave * 128 + 4 * l31) * 4u;
  f32x16 acc[RB][4];
  for (int r = 0; r < RB; ++r) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[r][j][e] = 0.f;
  f32x4 b[3][8];
  for (int s = 0; s < 3; ++s) for (int u = 0; u < 8; ++u) b[s][u] = f32x4{xin[tid + u], xin[tid + 64 + u], xin[tid + 128 + u], xin[tid + 192 + u]};
  float a[RB][8];
  for (int r = 0; r < RB; ++r) for (int u = 0; u < 8; ++u) a[r][u] = xin[(tid * 8 + u + 4096 * r) & 0xfffff];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < reps; ++r) {
    if (MODE & 2) {
      for (int u = 0; u < 8; ++u) b[0][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, voff, (2 * u) * 4096, 0));
      for (int u = 0; u < 8; ++u) b[1][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, voff, (2 * (8 + u)) * 4096, 0));
    }
#pragma unroll 1
    for (int blk = 0; blk < nblk; blk += 3) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int cur = blk + d;
        if (MODE & 2) {
          const int nb = cur + 2 < nblk ? cur + 2 : nblk - 1;
#pragma unroll
          for (int u = 0; u < 8; ++u)
            b[(d + 2) % 3][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, voff, (2 * (nb * 8 + u)) * 4096, 0));
        }
        if (MODE & 1) {
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            const float *xa = X + (32 * rb + l31) * 513 + 16 * (cur & 31) + kh;
#pragma unroll
            for (int u = 0; u < 8; ++u) a[rb][u] = xa[2 * u];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb][u], b[d][u][j], acc[rb][j], 0, 0, 0);
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int rb = 0; rb < RB; ++rb) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[rb][j][e];
  if (s == 12345.678f) out[tid] = s;
  if (lane == 0) { clk[2 * (blockIdx.x * 8 + wave)] = c1 - c0; clk[2 * (blockIdx.x * 8 + wave) + 1] = r1 - r0; }
}

template <int MODE, int RB>
void run(const char *name, const float *w, const float *xin, float *out, unsigned long long *clk) {
  const int wgs = 256, nblk = 33, reps = 1200 / RB;
  const size_t lds = (size_t)RB * 32 * 513 * 4;
  hipFuncSetAttribute((const void *)feed_kernel<MODE, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((feed_kernel<MODE, RB>), dim3(wgs), dim3(512), lds, 0, w, xin, out, clk, nblk, reps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((feed_kernel<MODE, RB>), dim3(wgs), dim3(512), lds, 0, w, xin, out, clk, nblk, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * wgs * 8);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double ghz = 0;
  for (int i = 0; i < wgs * 8; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
  ghz /= wgs * 8;
  const double flops = 3.0 * wgs * 8 * (double)reps * nblk * 32 * RB * 4096.0;
  printf("%-58s %8.2f ms  %7.1f TFLOP/s (%.3f of 157.3)  clock %.3f GHz  pipe busy %.3f\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3, ghz,
         flops / (ms * 1e-3) / (ghz * 1e9 * 1024 * 64));
}

int main() {
  float *w, *xin, *out; unsigned long long *clk;
  std::vector<float> h(1 << 20);
  srand(11); for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMalloc(&w, 1024 * 1024 * 4 + 65536); hipMemcpy(w, h.data(), 1 << 22, hipMemcpyHostToDevice);
  hipMalloc(&xin, 1 << 22); hipMemcpy(xin, h.data(), 1 << 22, hipMemcpyHostToDevice);
  hipMalloc(&out, 1 << 16); hipMalloc(&clk, 1 << 20);
  for (int pass = 0; pass < 2; ++pass) {
    run<0, 1>("registers only", w, xin, out, clk);
    run<1, 1>("+ A from LDS", w, xin, out, clk);
    run<2, 1>("+ B streamed from L2 (dwordx4, 2 blocks ahead)", w, xin, out, clk);
    run<3, 1>("A from LDS + B streamed (the kernel's K loop)", w, xin, out, clk);
    run<3, 2>("M = 64: two row blocks share every B fragment", w, xin, out, clk);
  }
  return 0;
}
