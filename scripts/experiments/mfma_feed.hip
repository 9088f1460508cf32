// micro-benchmark: what keeps v_mfma_f32_32x32x2_f32 from its peak in the K loop of mlp_group.hip?
// 256 workgroups x 8 waves, every wave owns 4 accumulator tiles and runs blocks of 32 MFMAs (8 k-steps x 4 tiles):
//   mode 0: registers only; 1: + A fragments from LDS (8 ds_read_b32 per block); 2: + B fragments streamed from memory
//   (8 buffer_load_dwordx4 per block, three blocks ahead, 2.6 MB array = L2 resident); 3: both (the kernel's K loop)
// hipcc --offload-arch=gfx950 -O3 -o mfma_feed mfma_feed.hip && ./mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void feed_kernel(const float *__restrict__ w, float *__restrict__ out, int nblk, int reps) {
  __shared__ float X[32 * 513];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  for (int i = tid; i < 32 * 513; i += 64 * NW) X[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, 0xffffffff, 0x00020000);
  const uint32_t voff = (uint32_t)(kh * 1024 + wave * 128 + 4 * l31) * 4u;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  f32x4 b[3][8];
  for (int s = 0; s < 3; ++s) for (int u = 0; u < 8; ++u) b[s][u] = f32x4{1.f, 2.f, 3.f, 4.f};
  float a[8];
  for (int u = 0; u < 8; ++u) a[u] = 1.0f + u;
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
          const float *xa = X + l31 * 513 + 16 * (cur & 31) + kh;
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] = xa[2 * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[d][u][j], acc[j], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  if (s == 12345.678f) out[tid] = s;
}

template <int MODE, int NW>
void run(const char *name, const float *w, float *out, int wgs) {
  const int nblk = 33, reps = 40;   // 33 blocks of 8 k-steps: the 512-deep third layer (+1 to make it a multiple of 3)
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  feed_kernel<MODE, NW><<<wgs, 64 * NW>>>(w, out, nblk, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  feed_kernel<MODE, NW><<<wgs, 64 * NW>>>(w, out, nblk, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)wgs * NW * reps * nblk * 32 * 4096.0;
  printf("%-44s waves/wg %d wgs %4d: %8.3f ms  %7.1f TFLOP/s (%.3f of 157.3)\n", name, NW, wgs, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
}

int main() {
  float *w, *out;
  hipMalloc(&w, 1024 * 1024 * 4 + 65536);
  hipMemset(w, 0, 1024 * 1024 * 4 + 65536);
  hipMalloc(&out, 1 << 16);
  for (int wgs : {256, 512}) {
    run<0, 8>("registers only", w, out, wgs);
    run<1, 8>("+ A from LDS", w, out, wgs);
    run<2, 8>("+ B streamed (dwordx4, 2 blocks ahead)", w, out, wgs);
    run<3, 8>("A from LDS + B streamed", w, out, wgs);
    run<0, 4>("registers only", w, out, wgs);
    run<3, 4>("A from LDS + B streamed", w, out, wgs);
  }
  return 0;
}
