// ball_query_grid.hip — grid-hashed two-shell radius search for the large SA layers (gfx950).
//
// Same results as two ball_query_{cnt,dilated}_kernel_fast launches
// (core/pcdet/ops/pointnet2/pointnet2_batch/src/ball_query_gpu.cu:53-130): the first `nsample` hits by
// ASCENDING POINT INDEX, cyclic padding, hit count — but a centre only looks at the points of the 3x3
// neighbouring columns of a uniform (x, y) grid whose cell edge is >= the outer radius, instead of
// sweeping all N points.
//
// The reference's "first nsample by index" contract forbids visiting candidates in spatial order, so
// the query is split in two exact steps:
//   1. every candidate of the 3x3 neighbourhood that passes a shell test sets bit k of a per-wave
//      N-bit LDS bitmap (ds_or_b32) — order-free;
//   2. the bitmap is read back in ascending word/bit order with a wave prefix-popcount, which yields
//      the hits sorted by index; the first nsample are kept.
// The distance arithmetic per (centre, point) pair is the same fma chain as everywhere else, so hit
// sets are bit-identical to the brute-force kernels'.
//
// Grid build: one workgroup per scene; cell counts / cursors live in LDS (<= 128 x 128 cells; when
// the scene is larger than 128 cells across, the cell edge grows instead — still >= the radius).
#include "common.h"

namespace {

constexpr int kGridMax = 128;                 // cells per axis
constexpr int kGridCells = kGridMax * kGridMax;
constexpr int kBuildThreads = 1024;
constexpr int kQueryWaves = 4;
constexpr int kMaxNs = 128;

struct GridHeader {   // per scene, 32 bytes
  float ox, oy, inv_cell;
  int nx, ny;
  int pad[3];
};

__device__ __forceinline__ float blk_reduce(float v, bool take_max, float *red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = take_max ? fmaxf(v, o) : fminf(v, o);
  }
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < kBuildThreads / 64; ++w) r = take_max ? fmaxf(r, red[w]) : fminf(r, red[w]);
  return r;
}

__device__ __forceinline__ int cell_coord(float p, float origin, float inv_cell, int ncell) {
  float f = (p - origin) * inv_cell;
  if (!(f >= 0.f)) f = 0.f;                       // NaN / below range -> first cell
  const float top = (float)(ncell - 1);
  if (f > top) f = top;
  return (int)f;
}

__global__ __launch_bounds__(kBuildThreads) void bq_grid_build_kernel(int n, float cell, const float *__restrict__ xyz,
                                                                      GridHeader *__restrict__ hdr,
                                                                      int *__restrict__ cell_start,
                                                                      float4 *__restrict__ sorted_pts) {
  __shared__ int counts[kGridCells];
  __shared__ float red[kBuildThreads / 64];
  __shared__ int wave_tot[kBuildThreads / 64];
  const int tid = threadIdx.x;
  xyz += (size_t)blockIdx.x * n * 3;
  cell_start += (size_t)blockIdx.x * (kGridCells + 1);
  sorted_pts += (size_t)blockIdx.x * n;   // (x, y, z, index bits) per sorted position: one 16-byte read per candidate

  float xmin = 3e38f, xmax = -3e38f, ymin = 3e38f, ymax = -3e38f;
  for (int k = tid; k < n; k += kBuildThreads) {
    const float x = xyz[(size_t)k * 3 + 0], y = xyz[(size_t)k * 3 + 1];
    if (fabsf(x) < 1e30f) { xmin = fminf(xmin, x); xmax = fmaxf(xmax, x); }
    if (fabsf(y) < 1e30f) { ymin = fminf(ymin, y); ymax = fmaxf(ymax, y); }
  }
  xmin = blk_reduce(xmin, false, red); xmax = blk_reduce(xmax, true, red);
  ymin = blk_reduce(ymin, false, red); ymax = blk_reduce(ymax, true, red);
  if (!(xmax >= xmin)) { xmin = 0.f; xmax = 0.f; }
  if (!(ymax >= ymin)) { ymin = 0.f; ymax = 0.f; }
  // cell edge: at least `cell` (the outer radius), larger if the scene would need > kGridMax cells
  float edge = cell;
  edge = fmaxf(edge, (xmax - xmin) / (float)(kGridMax - 1));
  edge = fmaxf(edge, (ymax - ymin) / (float)(kGridMax - 1));
  edge = edge * 1.05f + 1e-12f;                   // 5 % slack: rounding in (p - origin) * inv can never push
                                                  // two points closer than the radius two cells apart
  const float inv = 1.0f / edge;
  int nx = (int)((xmax - xmin) * inv) + 1, ny = (int)((ymax - ymin) * inv) + 1;
  nx = min(max(nx, 1), kGridMax);
  ny = min(max(ny, 1), kGridMax);
  const int ncells = nx * ny;
  if (tid == 0) {
    GridHeader h;
    h.ox = xmin; h.oy = ymin; h.inv_cell = inv; h.nx = nx; h.ny = ny; h.pad[0] = h.pad[1] = h.pad[2] = 0;
    hdr[blockIdx.x] = h;
  }
  for (int c = tid; c < ncells; c += kBuildThreads) counts[c] = 0;
  __syncthreads();
  for (int k = tid; k < n; k += kBuildThreads) {
    const int cx = cell_coord(xyz[(size_t)k * 3 + 0], xmin, inv, nx);
    const int cy = cell_coord(xyz[(size_t)k * 3 + 1], ymin, inv, ny);
    atomicAdd(&counts[cy * nx + cx], 1);
  }
  __syncthreads();
  // exclusive scan of counts[0..ncells): each thread owns a contiguous chunk
  const int chunk = (ncells + kBuildThreads - 1) / kBuildThreads;
  const int c0 = tid * chunk, c1 = min(c0 + chunk, ncells);
  int local = 0;
  for (int c = c0; c < c1; ++c) local += counts[c];
  int incl = local;
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  int run = base + incl - local;
  for (int c = c0; c < c1; ++c) {
    const int cnt = counts[c];
    cell_start[c] = run;
    counts[c] = run;                               // becomes the scatter cursor
    run += cnt;
  }
  if (tid == 0) cell_start[ncells] = n;
  __syncthreads();
  for (int k = tid; k < n; k += kBuildThreads) {
    const int cx = cell_coord(xyz[(size_t)k * 3 + 0], xmin, inv, nx);
    const int cy = cell_coord(xyz[(size_t)k * 3 + 1], ymin, inv, ny);
    const int pos = atomicAdd(&counts[cy * nx + cx], 1);
    sorted_pts[pos] = make_float4(xyz[(size_t)k * 3 + 0], xyz[(size_t)k * 3 + 1], xyz[(size_t)k * 3 + 2], __int_as_float(k));
  }
}

// one wave per centre; bitmaps of N bits per shell per wave in dynamic LDS
__global__ __launch_bounds__(64 * kQueryWaves) void bq_grid_query_kernel(
    int n, int m, int words, float rin2_a, float rout2_a, int ns_a, float rin2_b, float rout2_b, int ns_b,
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, const GridHeader *__restrict__ hdr,
    const int *__restrict__ cell_start, const float4 *__restrict__ sorted_pts, int *__restrict__ cnt_a,
    int *__restrict__ idx_a, int *__restrict__ cnt_b, int *__restrict__ idx_b) {
  extern __shared__ unsigned bitmaps[];            // [kQueryWaves][2][words]
  __shared__ int hits[kQueryWaves][2][kMaxNs];
  __shared__ int lists[kQueryWaves][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bs = blockIdx.y;
  unsigned *bm_a = bitmaps + (size_t)(wave * 2 + 0) * words;
  unsigned *bm_b = bitmaps + (size_t)(wave * 2 + 1) * words;
  for (int w = lane; w < words; w += 64) { bm_a[w] = 0u; bm_b[w] = 0u; }
  const GridHeader h = hdr[bs];
  const int *cs = cell_start + (size_t)bs * (kGridCells + 1);
  const float4 *si = sorted_pts + (size_t)bs * n;
  const int wpl = (words + 63) / 64;               // bitmap words per lane (contiguous block per lane)

  for (int ci = blockIdx.x * kQueryWaves + wave; ci < m; ci += gridDim.x * kQueryWaves) {
    const float *q = new_xyz + ((size_t)bs * m + ci) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const int cx = cell_coord(qx, h.ox, h.inv_cell, h.nx), cy = cell_coord(qy, h.oy, h.inv_cell, h.ny);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, h.nx - 1);
    const int y0 = max(cy - 1, 0), y1 = min(cy + 1, h.ny - 1);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // 1a. short lists: the hits of a centre (a handful on FPS-sampled clouds) are appended, unordered, to one 64-entry
    //     list per shell through ballots; each hit then finds its rank by index among the others (one lane per hit,
    //     an LDS broadcast read per other hit) — the "first nsample by ascending index" contract without touching the
    //     N-bit bitmaps.  A shell with more than 64 hits falls back to the bitmap path below for this centre.
    int na = 0, nb = 0;
    int *la = lists[wave][0], *lb = lists[wave][1];
    for (int y = y0; y <= y1; ++y) {
      const int beg = cs[y * h.nx + x0], end = cs[y * h.nx + x1 + 1];   // x-contiguous cells: one range
      for (int t0 = beg; t0 < end; t0 += 64) {
        const int t = t0 + lane;
        bool ha = false, hb = false;
        int k = 0;
        if (t < end) {
          const float4 c = si[t];
          k = __float_as_int(c.w);
          const float d2 = d6_sqdist(qx - c.x, qy - c.y, qz - c.z);
          ha = d2 >= rin2_a && d2 < rout2_a;
          hb = d2 >= rin2_b && d2 < rout2_b;
        }
        const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (ha) { const int pos = na + __popcll(ma & below); if (pos < 64) la[pos] = k; }
        if (hb) { const int pos = nb + __popcll(mb & below); if (pos < 64) lb[pos] = k; }
        na += __popcll(ma);
        nb += __popcll(mb);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    if (na <= 64 && nb <= 64) {
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) {
        const int *lst = sh == 0 ? la : lb;
        const int total = sh == 0 ? na : nb;
        const int ns = sh == 0 ? ns_a : ns_b;
        int *hbuf = hits[wave][sh];
        const int mine = lane < total ? lst[lane] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < total; ++j) rank += lst[j] < mine;      // point indices are distinct
        if (lane < total && rank < ns) hbuf[rank] = mine;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        const int cnt = min(total, ns);
        int *out = (sh == 0 ? idx_a : idx_b) + ((size_t)bs * m + ci) * ns;
        if (lane == 0) (sh == 0 ? cnt_a : cnt_b)[(size_t)bs * m + ci] = cnt;
        for (int l = lane; l < ns; l += 64) out[l] = cnt > 0 ? hbuf[l % cnt] : 0;
      }
      continue;
    }
    // 1b. (more than 64 hits in a shell) mark hits in the bitmaps (order-free)
    for (int y = y0; y <= y1; ++y) {
      const int beg = cs[y * h.nx + x0], end = cs[y * h.nx + x1 + 1];
      for (int t = beg + lane; t < end; t += 64) {
        const float4 c = si[t];
        const int k = __float_as_int(c.w);
        const float d2 = d6_sqdist(qx - c.x, qy - c.y, qz - c.z);
        if (d2 >= rin2_a && d2 < rout2_a) atomicOr(&bm_a[k >> 5], 1u << (k & 31));
        if (d2 >= rin2_b && d2 < rout2_b) atomicOr(&bm_b[k >> 5], 1u << (k & 31));
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // 2. read the bitmaps back in index order
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
      unsigned *bm = sh == 0 ? bm_a : bm_b;
      const int ns = sh == 0 ? ns_a : ns_b;
      int *hb = hits[wave][sh];
      const int w0 = lane * wpl, w1 = min(w0 + wpl, words);
      int mine = 0;
      for (int w = w0; w < w1; ++w) mine += __popc(bm[w]);
      int incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      const int total = __shfl(incl, 63);
      int rank = incl - mine;
      for (int w = w0; w < w1; ++w) {
        unsigned bits = bm[w];
        if (bits) bm[w] = 0u;                       // leave the bitmap clean for the next centre
        while (bits && rank < ns) {
          const int bit = __builtin_ctz(bits);
          bits &= bits - 1;
          hb[rank++] = (w << 5) + bit;
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
      const int cnt = min(total, ns);
      int *out = (sh == 0 ? idx_a : idx_b) + ((size_t)bs * m + ci) * ns;
      if (lane == 0) (sh == 0 ? cnt_a : cnt_b)[(size_t)bs * m + ci] = cnt;
      for (int l = lane; l < ns; l += 64) out[l] = cnt > 0 ? hb[l % cnt] : 0;
    }
  }
}

}  // namespace

DET6D_API int64_t det6d_ball_query_grid_workspace_bytes(int b, int n) {
  if (b <= 0 || n <= 0) return 0;
  const int64_t per = 32 + (int64_t)(kGridCells + 1) * 4 + (int64_t)n * 16;
  return (int64_t)b * ((per + 63) / 64 * 64) + 256;
}

DET6D_API int det6d_ball_query_pair_grid(int b, int n, int m, float rin_a, float rout_a, int ns_a, float rin_b,
                                         float rout_b, int ns_b, const float *new_xyz, const float *xyz,
                                         void *workspace, int *cnt_a, int *idx_a, int *cnt_b, int *idx_b,
                                         det6d_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0 || ns_a <= 0 || ns_b <= 0 || ns_a > kMaxNs || ns_b > kMaxNs || !new_xyz || !xyz ||
      !workspace || ((uintptr_t)workspace & 15) || !cnt_a || !idx_a || !cnt_b || !idx_b)
    return DET6D_EINVAL;
  const int words = (n + 31) / 32;
  const size_t lds = (size_t)kQueryWaves * 2 * words * sizeof(unsigned);
  if (lds > 96 * 1024) return DET6D_EINVAL;        // N <= 98304; use det6d_ball_query_pair beyond
  if (b == 0 || m == 0) return DET6D_OK;
  hipStream_t s = (hipStream_t)stream;
  // workspace layout: headers | cell_start | sorted (x, y, z, index) records
  char *ws = (char *)workspace;
  GridHeader *hdr = (GridHeader *)ws;
  size_t off = ((size_t)b * sizeof(GridHeader) + 63) / 64 * 64;
  int *cell_start = (int *)(ws + off);
  off += ((size_t)b * (kGridCells + 1) * 4 + 63) / 64 * 64;
  float4 *sorted_pts = (float4 *)(ws + off);
  const float rmax = rout_a > rout_b ? rout_a : rout_b;
  static bool big_lds = false;
  if (lds > 32 * 1024 && !big_lds) {
    hipError_t e = hipFuncSetAttribute((const void *)bq_grid_query_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       96 * 1024);
    if (e != hipSuccess) { det6d_set_error("det6d_ball_query_pair_grid hipFuncSetAttribute", e); return DET6D_ELAUNCH; }
    big_lds = true;
  }
  hipLaunchKernelGGL(bq_grid_build_kernel, dim3(b), dim3(kBuildThreads), 0, s, n, rmax, xyz, hdr, cell_start,
                     sorted_pts);
  const int blocks_x = min(det6d_divup(m, kQueryWaves), 1024);
  hipLaunchKernelGGL(bq_grid_query_kernel, dim3(blocks_x, b), dim3(64 * kQueryWaves), lds, s, n, m, words,
                     rin_a * rin_a, rout_a * rout_a, ns_a, rin_b * rin_b, rout_b * rout_b, ns_b, new_xyz, xyz, hdr,
                     cell_start, sorted_pts, cnt_a, idx_a, cnt_b, idx_b);
  return det6d_check_launch("det6d_ball_query_pair_grid");
}
