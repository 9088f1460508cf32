// fps.hip — farthest point sampling for gfx950 (wave64, DPP reductions, one barrier per round).
//
// Replaces farthest_point_sampling_kernel / furthest_point_sampling_weights_kernel of
// core/pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:101-222,419-540.
//
// Design (MI355X-first, not a translation of the CUDA kernel):
//  * one workgroup per scene; every point lives in VGPRs for the whole kernel (x,y,z,min-dist,
//    up to 16 points per lane at N=16384), so a round touches no memory except two 16-byte LDS
//    slots per wave;
//  * the reference resolves ties through the order of its strided scan + shared-memory halving
//    tree: among equal maxima the winner minimises (bitrev(k mod S), k), S = block size.  We get
//    exactly that order for free by giving hardware thread h the reference's virtual thread
//    v = bitrev(h): then "lowest hardware lane / lowest wave wins" is the reference's rule, and the
//    argmax becomes  wave max (6 DPP steps) -> ballot -> s_ff1 -> v_readlane;
//  * the winner's coordinates travel with the (value, index) pair through LDS, so the next round
//    starts without a dependent global load;
//  * LDS slots are double-buffered by round parity: one s_barrier per round.
#include "common.h"

namespace {

constexpr int kMaxPPT = 16;

__device__ __forceinline__ unsigned bitrev_n(unsigned v, int bits) {
  return bits == 0 ? 0u : (__builtin_bitreverse32(v) >> (32 - bits));
}

struct Slot {
  float val;
  int idx;
  float x, y, z;
  float pad[3];
};

// PPT points per thread in registers. WEIGHTED selects the S-FPS scoring rule.
template <int PPT, bool WEIGHTED>
__global__ __launch_bounds__(1024) void fps_reg_kernel(int n, int m, int log2s,
                                                       const float *__restrict__ xyz,
                                                       const float *__restrict__ weights,
                                                       float *__restrict__ temp,
                                                       int *__restrict__ idxs) {
  typedef float vecf __attribute__((ext_vector_type(PPT)));
  __shared__ Slot slots[2][16];

  const int S = 1 << log2s;
  const int h = threadIdx.x;
  const int lane = h & 63;
  const int wave = h >> 6;
  const int nwaves = (blockDim.x + 63) >> 6;
  const bool live = h < S;
  const int v = (int)bitrev_n((unsigned)h, log2s);

  xyz += (size_t)blockIdx.x * n * 3;
  temp += (size_t)blockIdx.x * n;
  idxs += (size_t)blockIdx.x * m;
  if (WEIGHTED) weights += (size_t)blockIdx.x * n;

  vecf px, py, pz, pt;
  double pw[PPT];
  float pwf[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = v + (j << log2s);
    const bool ok = live && k < n;
    px[j] = ok ? xyz[(size_t)k * 3 + 0] : 0.f;
    py[j] = ok ? xyz[(size_t)k * 3 + 1] : 0.f;
    pz[j] = ok ? xyz[(size_t)k * 3 + 2] : 0.f;
    pt[j] = ok ? temp[k] : 0.f;
    if (WEIGHTED) {
      pwf[j] = ok ? weights[k] : 0.f;
      // max(weights[k], 1e-12) with a double literal (sampling_gpu.cu:466); fmax drops a NaN weight
      pw[j] = fmax((double)pwf[j], 1e-12);
    }
  }

  float cx = 0.f, cy = 0.f, cz = 0.f;
  int first_round;
  if (WEIGHTED) {
    first_round = 0;
  } else {
    first_round = 1;
    if (h == 0) idxs[0] = 0;
    cx = xyz[0]; cy = xyz[1]; cz = xyz[2];
  }

  for (int r = first_round; r < m; ++r) {
    float best = live ? -1.0f : -__builtin_inff();
    int bj = 0;  // which of my points is my best (index into the register arrays)
    if (WEIGHTED && r == 0) {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const int k = v + (j << log2s);
        const float d = pwf[j];
        const bool up = live && k < n && d > best;
        bj = up ? j : bj;
        best = up ? d : best;
      }
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const int k = v + (j << log2s);
        const float d = d6_sqdist(px[j] - cx, py[j] - cy, pz[j] - cz);
        const float t = d6_fminf(d, pt[j]);
        pt[j] = t;
        float score = t;
        if (WEIGHTED) score = (float)((double)t * pw[j]);
        const bool up = live && k < n && score > best;
        bj = up ? j : bj;
        best = up ? score : best;
      }
    }
    // a live thread that found nothing keeps (-1, k=0) like the reference's (best=-1, besti=0)
    const bool found = best > -1.0f;
    const int bk = found ? v + (bj << log2s) : 0;

    // ---- wave argmax: lowest lane among the maxima wins ----
    const float wmax = d6_wave_max(best);
    const unsigned long long tie = __ballot(best == wmax);
    const int wl = __builtin_ctzll(tie);
    const int wj = d6_readlane_i(found ? bj : -1, wl);
    const int wk = d6_readlane_i(bk, wl);
    float sx, sy, sz;
    if (wj >= 0) {
      sx = d6_readlane_f(px[wj], wl);
      sy = d6_readlane_f(py[wj], wl);
      sz = d6_readlane_f(pz[wj], wl);
    } else {  // nothing beat -1: the reference then picks point 0
      sx = xyz[0]; sy = xyz[1]; sz = xyz[2];
    }
    Slot *sl = slots[r & 1];
    if (lane == 0) {
      sl[wave].val = wmax;
      sl[wave].idx = wk;
      sl[wave].x = sx; sl[wave].y = sy; sl[wave].z = sz;
    }
    __syncthreads();
    // ---- cross-wave argmax: lowest wave among the maxima wins ----
    const int src = lane < nwaves ? lane : 0;
    const float v2 = lane < nwaves ? sl[src].val : -__builtin_inff();
    const int i2 = sl[src].idx;
    const float x2 = sl[src].x, y2 = sl[src].y, z2 = sl[src].z;
    const float bmax = d6_wave_max(v2);
    const unsigned long long tie2 = __ballot(v2 == bmax);
    const int ww = __builtin_ctzll(tie2);
    const int old = d6_readlane_i(i2, ww);
    cx = d6_readlane_f(x2, ww);
    cy = d6_readlane_f(y2, ww);
    cz = d6_readlane_f(z2, ww);
    if (h == 0) idxs[r] = old;
  }

  // leave the final min-distances in temp like the reference does
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = v + (j << log2s);
    if (live && k < n) temp[k] = pt[j];
  }
}

// Generic fallback for N > 16 * 1024: same schedule, min-distances and coordinates stay in
// memory (L2) like the reference kernel.  Correct for any N; not a tuned path.
template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void fps_mem_kernel(int n, int m, int log2s,
                                                       const float *__restrict__ xyz,
                                                       const float *__restrict__ weights,
                                                       float *__restrict__ temp,
                                                       int *__restrict__ idxs) {
  __shared__ Slot slots[2][16];
  const int S = 1 << log2s;
  const int h = threadIdx.x;
  const int lane = h & 63;
  const int wave = h >> 6;
  const int nwaves = (blockDim.x + 63) >> 6;
  const bool live = h < S;
  const int v = (int)bitrev_n((unsigned)h, log2s);
  xyz += (size_t)blockIdx.x * n * 3;
  temp += (size_t)blockIdx.x * n;
  idxs += (size_t)blockIdx.x * m;
  if (WEIGHTED) weights += (size_t)blockIdx.x * n;

  int old = 0;
  int first_round = WEIGHTED ? 0 : 1;
  if (!WEIGHTED && h == 0) idxs[0] = 0;
  for (int r = first_round; r < m; ++r) {
    const float cx = xyz[(size_t)old * 3 + 0], cy = xyz[(size_t)old * 3 + 1], cz = xyz[(size_t)old * 3 + 2];
    float best = live ? -1.0f : -__builtin_inff();
    int bk = 0;
    if (live) {
      for (int k = v; k < n; k += S) {
        float score;
        if (WEIGHTED && r == 0) {
          score = weights[k];
        } else {
          const float d = d6_sqdist(xyz[(size_t)k * 3 + 0] - cx, xyz[(size_t)k * 3 + 1] - cy,
                                    xyz[(size_t)k * 3 + 2] - cz);
          const float t = d6_fminf(d, temp[k]);
          temp[k] = t;
          score = t;
          if (WEIGHTED) score = (float)((double)t * fmax((double)weights[k], 1e-12));
        }
        const bool up = score > best;
        bk = up ? k : bk;
        best = up ? score : best;
      }
    }
    const float wmax = d6_wave_max(best);
    const unsigned long long tie = __ballot(best == wmax);
    const int wl = __builtin_ctzll(tie);
    const int wk = d6_readlane_i(bk, wl);
    Slot *sl = slots[r & 1];
    if (lane == 0) { sl[wave].val = wmax; sl[wave].idx = wk; }
    __syncthreads();
    const int src = lane < nwaves ? lane : 0;
    const float v2 = lane < nwaves ? sl[src].val : -__builtin_inff();
    const int i2 = sl[src].idx;
    const float bmax = d6_wave_max(v2);
    const unsigned long long tie2 = __ballot(v2 == bmax);
    old = d6_readlane_i(i2, __builtin_ctzll(tie2));
    if (h == 0) idxs[r] = old;
  }
}

// core/pcdet/ops/pointnet2/pointnet2_batch/src/cuda_utils.h:10-14 (same double formula, same libm)
int opt_n_threads_log2(int work_size) {
  int pow_2 = (int)(log((double)work_size) / log(2.0));
  if (pow_2 > 10) pow_2 = 10;
  if (pow_2 < 0) pow_2 = 0;
  return pow_2;
}

template <bool W>
int launch_fps(int b, int n, int m, const float *xyz, const float *weights, float *temp, int *idx,
               hipStream_t stream) {
  if (b < 0 || n <= 0 || m < 0 || !xyz || !temp || !idx || (W && !weights)) return DET6D_EINVAL;
  if (b == 0 || m == 0) return DET6D_OK;
  const int log2s = opt_n_threads_log2(n);
  const int S = 1 << log2s;
  const int threads = S < 64 ? 64 : S;
  const int ppt = (n + S - 1) / S;
  dim3 grid(b), block(threads);
#define FPS_CASE(P)                                                                          \
  hipLaunchKernelGGL((fps_reg_kernel<P, W>), grid, block, 0, stream, n, m, log2s, xyz, weights, \
                     temp, idx)
  if (ppt <= 1) FPS_CASE(1);
  else if (ppt <= 2) FPS_CASE(2);
  else if (ppt <= 4) FPS_CASE(4);
  else if (ppt <= 8) FPS_CASE(8);
  else if (ppt <= kMaxPPT) FPS_CASE(16);
  else
    hipLaunchKernelGGL((fps_mem_kernel<W>), grid, block, 0, stream, n, m, log2s, xyz, weights, temp,
                       idx);
#undef FPS_CASE
  return det6d_check_launch("det6d_fps");
}

}  // namespace

DET6D_API int det6d_fps(int b, int n, int m, const float *xyz, float *temp, int *idx,
                        det6d_stream_t stream) {
  return launch_fps<false>(b, n, m, xyz, nullptr, temp, idx, (hipStream_t)stream);
}

DET6D_API int det6d_fps_weights(int b, int n, int m, const float *xyz, const float *weights,
                                float *temp, int *idx, det6d_stream_t stream) {
  return launch_fps<true>(b, n, m, xyz, weights, temp, idx, (hipStream_t)stream);
}
