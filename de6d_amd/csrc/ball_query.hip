// ball_query.hip — radius neighbour search for gfx950: one wave64 per centre, lanes sweep 64
// consecutive candidate points per step (coalesced), hits are appended in ascending point index
// with __ballot + prefix popcount, the wave leaves as soon as nsample hits are in.
//
// Replaces ball_query_kernel_fast / ball_query_cnt_kernel_fast / ball_query_dilated_kernel_fast
// (core/pcdet/ops/pointnet2/pointnet2_batch/src/ball_query_gpu.cu:15-130), which give one thread
// per centre a serial scan over all N points.  Result semantics are identical: first `nsample`
// hits by ascending index; cnt/dilated variants repeat the hit list cyclically and report the
// count; the plain variant pads with the first hit; zero hits leave `idx` untouched.
#include "common.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kMaxSample = 128;

enum { BQ_PLAIN = 0, BQ_CNT = 1, BQ_DILATED = 2 };

template <int MODE>
__global__ __launch_bounds__(64 * kWavesPerBlock) void ball_query_kernel(
    int n, int m, float rin2, float rout2, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx_cnt, int *__restrict__ idx) {
  __shared__ int hits[kWavesPerBlock][kMaxSample];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int bs = blockIdx.y;
  const int pt = blockIdx.x * kWavesPerBlock + wave;
  if (pt >= m) return;  // whole wave leaves together

  const float *q = new_xyz + ((size_t)bs * m + pt) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  const float *p = xyz + (size_t)bs * n * 3;
  int *hb = hits[wave];

  int cnt = 0;
  for (int k0 = 0; k0 < n; k0 += 64) {
    const int k = k0 + lane;
    bool hit = false;
    if (k < n) {
      const float x = p[(size_t)k * 3 + 0], y = p[(size_t)k * 3 + 1], z = p[(size_t)k * 3 + 2];
      const float d2 = d6_sqdist(qx - x, qy - y, qz - z);
      hit = (MODE == BQ_DILATED) ? (d2 >= rin2 && d2 < rout2) : (d2 < rout2);
    }
    const unsigned long long mask = __ballot(hit);
    if (mask) {
      const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      if (hit && pos < nsample) hb[pos] = k;
      cnt += __popcll(mask);
      if (cnt >= nsample) break;
    }
  }
  if (cnt > nsample) cnt = nsample;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the LDS appends above are visible wave-wide

  int *out = idx + ((size_t)bs * m + pt) * nsample;
  if (MODE != BQ_PLAIN && lane == 0) idx_cnt[(size_t)bs * m + pt] = cnt;
  if (cnt > 0) {
    for (int l = lane; l < nsample; l += 64) {
      int v;
      if (MODE == BQ_PLAIN) v = l < cnt ? hb[l] : hb[0];  // ball_query_gpu.cu:41-47
      else v = hb[l % cnt];                                // ball_query_gpu.cu:86-89,126-129
      out[l] = v;
    }
  }
}

template <int MODE>
int launch(int b, int n, int m, float rin, float rout, int nsample, const float *new_xyz,
           const float *xyz, int *idx_cnt, int *idx, hipStream_t stream) {
  if (b < 0 || n < 0 || m < 0 || nsample <= 0 || nsample > kMaxSample || !new_xyz || !xyz || !idx ||
      (MODE != BQ_PLAIN && !idx_cnt))
    return DET6D_EINVAL;
  if (b == 0 || m == 0) return DET6D_OK;
  const float rin2 = rin * rin, rout2 = rout * rout;  // fp32 products like the reference
  dim3 grid(det6d_divup(m, kWavesPerBlock), b), block(64 * kWavesPerBlock);
  hipLaunchKernelGGL((ball_query_kernel<MODE>), grid, block, 0, stream, n, m, rin2, rout2, nsample,
                     new_xyz, xyz, idx_cnt, idx);
  return det6d_check_launch("det6d_ball_query");
}

}  // namespace

DET6D_API int det6d_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                               const float *xyz, int *idx, det6d_stream_t stream) {
  return launch<BQ_PLAIN>(b, n, m, 0.f, radius, nsample, new_xyz, xyz, nullptr, idx, (hipStream_t)stream);
}

DET6D_API int det6d_ball_query_cnt(int b, int n, int m, float radius, int nsample,
                                   const float *new_xyz, const float *xyz, int *idx_cnt, int *idx,
                                   det6d_stream_t stream) {
  return launch<BQ_CNT>(b, n, m, 0.f, radius, nsample, new_xyz, xyz, idx_cnt, idx, (hipStream_t)stream);
}

DET6D_API int det6d_ball_query_dilated(int b, int n, int m, float radius_in, float radius_out,
                                       int nsample, const float *new_xyz, const float *xyz,
                                       int *idx_cnt, int *idx, det6d_stream_t stream) {
  return launch<BQ_DILATED>(b, n, m, radius_in, radius_out, nsample, new_xyz, xyz, idx_cnt, idx,
                            (hipStream_t)stream);
}
