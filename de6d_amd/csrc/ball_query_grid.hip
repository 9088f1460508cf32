// ball_query_grid.hip — grid-hashed two-shell radius search for the large SA layers (gfx950).
//
// Same results as two ball_query_{cnt,dilated}_kernel_fast launches
// (core/pcdet/ops/pointnet2/pointnet2_batch/src/ball_query_gpu.cu:53-130): the first `nsample` hits by
// ASCENDING POINT INDEX, cyclic padding, hit count — but a centre only looks at the points of the 3x3
// neighbouring columns of a uniform (x, y) grid whose cell edge is >= the outer radius, instead of
// sweeping all N points.
//
// The reference's "first nsample by index" contract forbids taking candidates in spatial order: per shell the
// query keeps a running selection of the nsample smallest hit indices in a short LDS list with a pruning
// threshold (bq_grid_query_kernel below) and ranks it once at the end.
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
constexpr int kMaxNs = 64;                    // nsample of a shell: one list entry per lane when the list is ranked

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

// Keeps the min(n, ns) SMALLEST entries of lst[0..n) (n <= kListCap, distinct point indices), ascending, in lst[0..); returns
// how many.  One or two entries per lane, each finds its rank among the others (LDS broadcast reads).
__device__ __forceinline__ int keep_smallest(int *__restrict__ lst, int *__restrict__ tmp, int n, int ns, int lane) {
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  const int e0 = lane < n ? lst[lane] : 0x7fffffff;
  const int e1 = lane + 64 < n ? lst[lane + 64] : 0x7fffffff;
  int r0 = 0, r1 = 0;
  if (n <= 64) {
    for (int j = 0; j < n; ++j) r0 += lst[j] < e0;
  } else {
    for (int j = 0; j < n; ++j) { const int v = lst[j]; r0 += v < e0; r1 += v < e1; }
  }
  if (lane < n && r0 < ns) tmp[r0] = e0;
  if (lane + 64 < n && r1 < ns) tmp[r1] = e1;
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  const int keep = n < ns ? n : ns;
  if (lane < keep) lst[lane] = tmp[lane];          // ns <= 64
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  return keep;
}

// One wave per centre.  The reference's contract is "the first nsample hits by ASCENDING POINT INDEX", but the grid hands
// out candidates in spatial order, so per shell the wave keeps a running selection of the nsample smallest hit indices:
// hits are appended, unordered, to a kListCap-entry LDS list through ballots; when the list would overflow it is cut back
// to its nsample smallest entries and the largest of them becomes a THRESHOLD — a later hit with a larger index can never
// be among the first nsample and is dropped before it is stored.  After the first cut the list holds >= nsample entries for
// good, so the hit count min(total, nsample) is known without counting the dropped hits.  On FPS-sampled clouds a shell has
// a handful of hits and the list is ranked once, at the end; in the dense parts of a real sweep (hundreds of hits in the
// 0.8 m shell of the first SA layer) a cut happens once or twice per centre.  (Round 2 re-scanned such centres into an
// N-bit LDS bitmap per wave and read it back in index order: 273 us for SA1 on ray-cast scenes against 31 us on uniform
// ones; bit-identical results.)
constexpr int kListCap = 128;

__global__ __launch_bounds__(64 * kQueryWaves) void bq_grid_query_kernel(
    int n, int m, float rin2_a, float rout2_a, int ns_a, float rin2_b, float rout2_b, int ns_b,
    const float *__restrict__ new_xyz, const GridHeader *__restrict__ hdr,
    const int *__restrict__ cell_start, const float4 *__restrict__ sorted_pts, int *__restrict__ cnt_a,
    int *__restrict__ idx_a, int *__restrict__ cnt_b, int *__restrict__ idx_b) {
  __shared__ int lists[kQueryWaves][2][kListCap];
  __shared__ int tmp[kQueryWaves][kMaxNs];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bs = blockIdx.y;
  const GridHeader h = hdr[bs];
  const int *cs = cell_start + (size_t)bs * (kGridCells + 1);
  const float4 *si = sorted_pts + (size_t)bs * n;
  int *la = lists[wave][0], *lb = lists[wave][1], *tw = tmp[wave];
  const unsigned long long below = (1ull << lane) - 1ull;

  for (int ci = blockIdx.x * kQueryWaves + wave; ci < m; ci += gridDim.x * kQueryWaves) {
    const float *q = new_xyz + ((size_t)bs * m + ci) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const int cx = cell_coord(qx, h.ox, h.inv_cell, h.nx), cy = cell_coord(qy, h.oy, h.inv_cell, h.ny);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, h.nx - 1);
    const int y0 = max(cy - 1, 0), y1 = min(cy + 1, h.ny - 1);
    int na = 0, nb = 0;                              // entries in the lists (wave-uniform)
    int thr_a = 0x7fffffff, thr_b = 0x7fffffff;      // only hits with a smaller point index can still be among the first nsample
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
          ha = d2 >= rin2_a && d2 < rout2_a && k < thr_a;
          hb = d2 >= rin2_b && d2 < rout2_b && k < thr_b;
        }
        unsigned long long ma = __ballot(ha), mb = __ballot(hb);
        if (na + __popcll(ma) > kListCap) {          // wave-uniform: cut the list back to its ns_a smallest entries
          na = keep_smallest(la, tw, na, ns_a, lane);
          thr_a = la[ns_a - 1];                      // na == ns_a here (the list held more than 64 >= ns_a entries)
          ha = ha && k < thr_a;
          ma = __ballot(ha);
        }
        if (nb + __popcll(mb) > kListCap) {
          nb = keep_smallest(lb, tw, nb, ns_b, lane);
          thr_b = lb[ns_b - 1];
          hb = hb && k < thr_b;
          mb = __ballot(hb);
        }
        if (ha) la[na + __popcll(ma & below)] = k;
        if (hb) lb[nb + __popcll(mb & below)] = k;
        na += __popcll(ma);
        nb += __popcll(mb);
      }
    }
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
      int *lst = sh == 0 ? la : lb;
      const int ns = sh == 0 ? ns_a : ns_b;
      const int cnt = keep_smallest(lst, tw, sh == 0 ? na : nb, ns, lane);   // = min(total hits, ns): see the header
      int *out = (sh == 0 ? idx_a : idx_b) + ((size_t)bs * m + ci) * ns;
      if (lane == 0) (sh == 0 ? cnt_a : cnt_b)[(size_t)bs * m + ci] = cnt;
      for (int l = lane; l < ns; l += 64) out[l] = cnt > 0 ? lst[l % cnt] : 0;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();                 // the lists are rewritten by the next centre
  }
}

}  // namespace

// largest scene the per-scene build kernel bins (LDS histogram of kGridCells cells, one workgroup per scene)
constexpr int kGridMaxN = 98304;

DET6D_API int det6d_ball_query_grid_supported(int n, int ns_a, int ns_b) {
  return n > 0 && n <= kGridMaxN && ns_a > 0 && ns_b > 0 && ns_a <= kMaxNs && ns_b <= kMaxNs;
}

DET6D_API int64_t det6d_ball_query_grid_workspace_bytes(int b, int n) {
  if (b <= 0 || n <= 0) return 0;
  const int64_t per = 32 + (int64_t)(kGridCells + 1) * 4 + (int64_t)n * 16;
  return (int64_t)b * ((per + 63) / 64 * 64) + 256;
}

DET6D_API int det6d_ball_query_pair_grid(int b, int n, int m, float rin_a, float rout_a, int ns_a, float rin_b,
                                         float rout_b, int ns_b, const float *new_xyz, const float *xyz,
                                         void *workspace, int *cnt_a, int *idx_a, int *cnt_b, int *idx_b,
                                         det6d_stream_t stream) {
  if (b < 0 || m < 0 || !det6d_ball_query_grid_supported(n, ns_a, ns_b) || !new_xyz || !xyz ||
      !workspace || ((uintptr_t)workspace & 15) || !cnt_a || !idx_a || !cnt_b || !idx_b)
    return DET6D_EINVAL;
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
  hipLaunchKernelGGL(bq_grid_build_kernel, dim3(b), dim3(kBuildThreads), 0, s, n, rmax, xyz, hdr, cell_start,
                     sorted_pts);
  const int blocks_x = min(det6d_divup(m, kQueryWaves), 1024);
  hipLaunchKernelGGL(bq_grid_query_kernel, dim3(blocks_x, b), dim3(64 * kQueryWaves), 0, s, n, m,
                     rin_a * rin_a, rout_a * rout_a, ns_a, rin_b * rin_b, rout_b * rout_b, ns_b, new_xyz, hdr,
                     cell_start, sorted_pts, cnt_a, idx_a, cnt_b, idx_b);
  return det6d_check_launch("det6d_ball_query_pair_grid");
}
