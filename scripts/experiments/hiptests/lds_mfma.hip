// micro-benchmark: the GEMM inner loop (operands from LDS, 2x2 MFMA tiles per wave, BK = 16) with
//  V0: k-major tiles read by ds_read_b32 pairs (what linear.hip does)
//  V1: per-lane-half k-contiguous tiles [kh][row][8] (XOR-swizzled 16-B slots) read by ds_read_b128
//  V2: A as V1, B as V0
// No global loads, no barriers inside the loop: isolates LDS operand delivery + MFMA issue.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, BK = 16;

template <int VARIANT>
__global__ __launch_bounds__(256) void k(float *out, int slabs) {
  __shared__ float As[BK * (BM + 2)];
  __shared__ float Bs[BK * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < BK * (BM + 2); i += 256) As[i] = 1e-3f * (i % 97);
  for (int i = tid; i < BK * BN; i += 256) Bs[i] = 1e-3f * (i % 89);
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  for (int s = 0; s < slabs; ++s) {
    if (VARIANT == 0) {
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        float af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = As[(2 * ks + kh) * (BM + 2) + wm * 64 + l31 + 32 * i];
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = Bs[(2 * ks + kh) * BN + wn * 64 + l31 + 32 * j];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else if (VARIANT == 2) {       // A by ds_read_b128 (k-contiguous per lane half), B k-major by ds_read_b32
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 a4[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wm * 64 + l31 + 32 * i;
          a4[i] = *reinterpret_cast<const float4 *>(&As[((kh * BM + row) * 2 + (h ^ ((row >> 3) & 1))) * 4]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int ks = 4 * h + t;
          float bf[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[j] = Bs[(2 * ks + kh) * BN + wn * 64 + l31 + 32 * j];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const float a = t == 0 ? a4[i].x : t == 1 ? a4[i].y : t == 2 ? a4[i].z : a4[i].w;
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[j], acc[i][j], 0, 0, 0);
            }
        }
      }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {   // 4 k-steps per 16-B slot
        float4 a4[2], b4[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wm * 64 + l31 + 32 * i;
          a4[i] = *reinterpret_cast<const float4 *>(&As[((kh * BM + row) * 2 + (h ^ ((row >> 3) & 1))) * 4]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = wn * 64 + l31 + 32 * j;
          b4[j] = *reinterpret_cast<const float4 *>(&Bs[((kh * BN + col) * 2 + (h ^ ((col >> 3) & 1))) * 4]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const float a = t == 0 ? a4[i].x : t == 1 ? a4[i].y : t == 2 ? a4[i].z : a4[i].w;
              const float b = t == 0 ? b4[j].x : t == 1 ? b4[j].y : t == 2 ? b4[j].z : b4[j].w;
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
            }
      }
    }
  }
  float sum = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
  out[blockIdx.x * 256 + tid] = sum;
}

template <int VARIANT>
void run(int blocks_per_cu, int slabs = 400) {
  float *d;
  const int blocks = 256 * blocks_per_cu;
  hipMalloc(&d, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<VARIANT>, dim3(blocks), dim3(256), 0, 0, d, slabs);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<VARIANT>, dim3(blocks), dim3(256), 0, 0, d, slabs);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * slabs * 32 * 4096.0;
  printf("variant %d, %d blocks/CU (= waves/SIMD), %4d slabs, %.2f ms: %.1f TF\n", VARIANT, blocks_per_cu, slabs, ms, flops / ms / 1e9);
  hipFree(d);
}

int main() {
  for (int b = 1; b <= 3; ++b) run<0>(b);
  for (int b = 1; b <= 3; ++b) run<1>(b);
  for (int b = 3; b <= 4; ++b) run<2>(b);
  run<0>(4); run<1>(4);
  // duration sweep at 3 waves/SIMD: separates occupancy from clock management under sustained matrix load
  run<0>(3, 100); run<0>(3, 400); run<0>(3, 800); run<0>(3, 1600); run<0>(3, 6400); run<0>(4, 100); run<0>(4, 300); run<0>(2, 600);
  return 0;
}
