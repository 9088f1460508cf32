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
constexpr int kMaxSample = 128;      // fused two-shell kernel: static hit lists
constexpr int kMaxSampleDyn = 4096;  // single-shell kernels: one dynamic-LDS hit list of nsample entries per wave (64 KB)

enum { BQ_PLAIN = 0, BQ_CNT = 1, BQ_DILATED = 2 };

template <int MODE>
__global__ __launch_bounds__(64 * kWavesPerBlock) void ball_query_kernel(
    int n, int m, float rin2, float rout2, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx_cnt, int *__restrict__ idx) {
  extern __shared__ int hits[];      // [kWavesPerBlock][nsample]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int bs = blockIdx.y;
  const int pt = blockIdx.x * kWavesPerBlock + wave;
  if (pt >= m) return;  // whole wave leaves together

  const float *q = new_xyz + ((size_t)bs * m + pt) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  const float *p = xyz + (size_t)bs * n * 3;
  int *hb = hits + wave * nsample;

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
  if (b < 0 || n < 0 || m < 0 || nsample <= 0 || nsample > kMaxSampleDyn || !new_xyz || !xyz || !idx ||
      (MODE != BQ_PLAIN && !idx_cnt))
    return DET6D_EINVAL;
  if (b == 0 || m == 0) return DET6D_OK;
  const float rin2 = rin * rin, rout2 = rout * rout;  // fp32 products like the reference
  dim3 grid(det6d_divup(m, kWavesPerBlock), b), block(64 * kWavesPerBlock);
  hipLaunchKernelGGL((ball_query_kernel<MODE>), grid, block, (size_t)kWavesPerBlock * nsample * sizeof(int), stream, n, m, rin2, rout2, nsample,
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

// ------------------------------------------------------------------------------------------------
// Fused two-shell query: the two radius groups of an SA layer share centres and candidates, so one
// sweep evaluates every (centre, point) distance once and feeds both hit lists.  A wave owns four
// centres (two packed pairs: v_pk_add/mul/fma_f32 evaluate two centres per instruction), so the
// three coordinate loads of a candidate are amortised over eight shell tests.
// Shell g accepts  rin2_g <= d2 < rout2_g  (rin = 0 gives the plain ball of ball_query_cnt).
// Same result contract as the single-shell kernels, except that empty balls are written as zeros
// here (the single-shell entry points leave the caller's zero fill untouched), so callers need
// no memset.
// ------------------------------------------------------------------------------------------------
namespace {

typedef float bq_f32x2 __attribute__((ext_vector_type(2)));
constexpr int kPairCentres = 4;   // centres per wave
constexpr int kPairWaves = 4;

__global__ __launch_bounds__(64 * kPairWaves) void ball_query_pair_kernel(
    int n, int m, float rin2_a, float rout2_a, int ns_a, float rin2_b, float rout2_b, int ns_b,
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, int *__restrict__ cnt_a,
    int *__restrict__ idx_a, int *__restrict__ cnt_b, int *__restrict__ idx_b) {
  __shared__ int hits[kPairWaves][kPairCentres][2][kMaxSample];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int bs = blockIdx.y;
  const int c0 = (blockIdx.x * kPairWaves + wave) * kPairCentres;
  if (c0 >= m) return;
  const float *p = xyz + (size_t)bs * n * 3;

  bq_f32x2 qx[2], qy[2], qz[2];
#pragma unroll
  for (int c = 0; c < kPairCentres; ++c) {
    const int ci = min(c0 + c, m - 1);  // tail centres duplicate the last one (never stored)
    const float *q = new_xyz + ((size_t)bs * m + ci) * 3;
    qx[c >> 1][c & 1] = q[0];
    qy[c >> 1][c & 1] = q[1];
    qz[c >> 1][c & 1] = q[2];
  }
  int ca[kPairCentres] = {0, 0, 0, 0}, cb[kPairCentres] = {0, 0, 0, 0};

  for (int k0 = 0; k0 < n; k0 += 64) {
    const int k = k0 + lane;
    const bool ok = k < n;
    const int kk = ok ? k : n - 1;
    const float x = p[(size_t)kk * 3 + 0], y = p[(size_t)kk * 3 + 1], z = p[(size_t)kk * 3 + 2];
    const bq_f32x2 x2 = {x, x}, y2 = {y, y}, z2 = {z, z};
    bool all_done = true;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const bq_f32x2 dx = qx[pr] - x2, dy = qy[pr] - y2, dz = qz[pr] - z2;
      bq_f32x2 d = dy * dy;
      d = __builtin_elementwise_fma(dx, dx, d);
      d = __builtin_elementwise_fma(dz, dz, d);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = 2 * pr + e;
        const float d2 = d[e];
        const unsigned long long ma = __ballot(ok && d2 >= rin2_a && d2 < rout2_a);
        const unsigned long long mb = __ballot(ok && d2 >= rin2_b && d2 < rout2_b);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (ma && ca[c] < ns_a) {
          const int pos = ca[c] + __popcll(ma & below);
          if (((ma >> lane) & 1ull) && pos < ns_a) hits[wave][c][0][pos] = k;
          ca[c] += __popcll(ma);
        }
        if (mb && cb[c] < ns_b) {
          const int pos = cb[c] + __popcll(mb & below);
          if (((mb >> lane) & 1ull) && pos < ns_b) hits[wave][c][1][pos] = k;
          cb[c] += __popcll(mb);
        }
        all_done = all_done && ca[c] >= ns_a && cb[c] >= ns_b;
      }
    }
    if (all_done) break;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();

#pragma unroll
  for (int c = 0; c < kPairCentres; ++c) {
    const int ci = c0 + c;
    if (ci >= m) break;
    const int na = min(ca[c], ns_a), nb = min(cb[c], ns_b);
    if (lane == 0) {
      cnt_a[(size_t)bs * m + ci] = na;
      cnt_b[(size_t)bs * m + ci] = nb;
    }
    int *oa = idx_a + ((size_t)bs * m + ci) * ns_a;
    int *ob = idx_b + ((size_t)bs * m + ci) * ns_b;
    for (int l = lane; l < ns_a; l += 64) oa[l] = na > 0 ? hits[wave][c][0][l % na] : 0;
    for (int l = lane; l < ns_b; l += 64) ob[l] = nb > 0 ? hits[wave][c][1][l % nb] : 0;
  }
}

}  // namespace

DET6D_API int det6d_ball_query_pair(int b, int n, int m, float rin_a, float rout_a, int ns_a, float rin_b,
                                    float rout_b, int ns_b, const float *new_xyz, const float *xyz,
                                    int *cnt_a, int *idx_a, int *cnt_b, int *idx_b, det6d_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0 || ns_a <= 0 || ns_b <= 0 || ns_a > kMaxSample || ns_b > kMaxSample || !new_xyz ||
      !xyz || !cnt_a || !idx_a || !cnt_b || !idx_b)
    return DET6D_EINVAL;
  if (b == 0 || m == 0) return DET6D_OK;
  dim3 grid(det6d_divup(m, kPairWaves * kPairCentres), b), block(64 * kPairWaves);
  hipLaunchKernelGGL(ball_query_pair_kernel, grid, block, 0, (hipStream_t)stream, n, m, rin_a * rin_a,
                     rout_a * rout_a, ns_a, rin_b * rin_b, rout_b * rout_b, ns_b, new_xyz, xyz, cnt_a, idx_a,
                     cnt_b, idx_b);
  return det6d_check_launch("det6d_ball_query_pair");
}
