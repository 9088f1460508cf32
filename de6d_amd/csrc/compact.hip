// compact.hip — ragged ("compact") row lists for the grouped SA MLPs.
//
// The reference materialises nsample rows per centre and pads a ball that holds cnt < nsample points by
// repeating its first cnt hits (ball_query_gpu.cu:75-90,114-129: `for l = 0; cnt < nsample; ++l, ++cnt:
// idx[cnt] = idx[l]`); the grouped MLP (pointnet2_modules.py:462-467) is a pointwise function of a row
// followed by a max over the nsample rows.  Rows >= cnt are therefore exact duplicates of rows < cnt, and
//     max over nsample rows  ==  max over the first s rows        for every s with cnt <= s <= nsample,
// bit for bit.  On FPS-sampled clouds most balls are far from full (mean cnt 1..10 of 16 / 32), so the MLP
// is evaluated on the first s = 2^ceil(log2(cnt)) slots of each centre only (s >= smin; empty balls take
// the smallest class: their rows are computed from the reference's idx = 0 and masked to zero as before).
//
// Row space: centres are binned by class s in {32, 16, 8, 4, 2, 1}; each class owns one contiguous region
// of the compact row space, the classes in descending s, every region padded to a multiple of 128 rows
// (one GEMM row tile holds a single class), centres in ascending order inside a region.
//   hdr[0]      total rows (multiple of 128): the GEMMs read their row count HERE, on the device
//   hdr[1 + c]  end of the region of class c (s = 32 >> c), c = 0..5;  hdr[6] == hdr[0]
//   hdr[8]      sum of min(cnt, ns) (rows that carry information), hdr[9] rows before the 128-row alignment
//   crow_p[r]   global point row (scene * n + neighbour index) row r gathers
//   crow_c[r]   centre (scene * m + j) row r belongs to, bit 30 set when its ball is empty (pooled value = 0:
//               pointnet2_modules.py:465-467), -1 for alignment rows (computed, never stored)
#include "common.h"
#include "compact_parts.h"

namespace {

constexpr int kClasses = kCompactClasses;

__device__ __forceinline__ int parts_of(int cnt, int ns, int smin, int split_tol, int *rows_out) {
  return d6_compact_parts_of(cnt, ns, smin, split_tol, rows_out);
}

// pass 1: parts per class (and information rows) of every block of 256 centres -> table[block][kClasses + 1]
// one radius group of a layer: both groups of an SA layer share total / n / m and go through ONE launch (grid.y)
struct GroupArgs {
  int ns, smin, split;
  const int *cnt, *idx;
  int *hdr, *crow_p, *crow_c;
  int col0, width;
};
struct PairArgs { GroupArgs g[2]; };

__global__ __launch_bounds__(256) void compact_count_kernel(int total, const PairArgs pa) {
  const GroupArgs &ga = pa.g[blockIdx.y];
  const int ns = ga.ns, smin = ga.smin, split = ga.split;
  const int *__restrict__ cnt = ga.cnt;
  int *__restrict__ table = ga.hdr + 16;
  __shared__ int acc[kClasses + 1];
  const int tid = threadIdx.x, i = blockIdx.x * 256 + tid;
  if (tid <= kClasses) acc[tid] = 0;
  __syncthreads();
  int rows = 0, k = 0;
  const int mask = i < total ? parts_of(k = cnt[i], ns, smin, split, &rows) : 0;
  const int real = i < total ? (k < 0 ? 0 : (k < ns ? k : ns)) : 0;
#pragma unroll
  for (int cc = 0; cc < kClasses; ++cc) {
    const int c = __popcll(__ballot((mask >> cc) & 1));
    if ((tid & 63) == 0 && c) atomicAdd(&acc[cc], c);
  }
  int r = real;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o);
  if ((tid & 63) == 0 && r) atomicAdd(&acc[kClasses], r);
  __syncthreads();
  if (tid <= kClasses) table[blockIdx.x * (kClasses + 1) + tid] = acc[tid];
}

// pass 2: one workgroup per block of 256 centres: region starts and the block's first row per class from the table,
// then ordered placement (ballot ranks per class), one thread per (centre, slot) for the row writes
__global__ __launch_bounds__(256) void compact_place_kernel(int total, int n, int m, int nblk, float *__restrict__ zero_y, int ldy,
                                                            const PairArgs pa) {
  const GroupArgs &ga = pa.g[blockIdx.y];
  const int ns = ga.ns, smin = ga.smin, split = ga.split, col0 = ga.col0, width = ga.width;
  const int *__restrict__ cnt = ga.cnt;
  const int *__restrict__ idx = ga.idx;
  int *__restrict__ hdr = ga.hdr;
  const int *__restrict__ table = ga.hdr + 16;
  int *__restrict__ crow_p = ga.crow_p;
  int *__restrict__ crow_c = ga.crow_c;
  __shared__ int h_all[kClasses + 1], h_before[kClasses];
  __shared__ int wave_cnt[4][kClasses];
  __shared__ int start[kClasses + 1], base[kClasses];
  __shared__ int r0_s[256][kClasses];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid <= kClasses) h_all[tid] = 0;
  if (tid < kClasses) h_before[tid] = 0;
  __syncthreads();
  {
    int la[kClasses + 1], lb[kClasses];
#pragma unroll
    for (int c = 0; c <= kClasses; ++c) la[c] = 0;
#pragma unroll
    for (int c = 0; c < kClasses; ++c) lb[c] = 0;
    for (int bk = tid; bk < nblk; bk += 256) {
#pragma unroll
      for (int c = 0; c <= kClasses; ++c) {
        const int v = table[bk * (kClasses + 1) + c];
        la[c] += v;
        if (c < kClasses && bk < (int)blockIdx.x) lb[c] += v;
      }
    }
    // wave sums first: 256 threads hammering 13 LDS words with atomics serialise (~15 us of this kernel)
#pragma unroll
    for (int c = 0; c <= kClasses; ++c) {
      int v = la[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0 && v) atomicAdd(&h_all[c], v);
    }
#pragma unroll
    for (int c = 0; c < kClasses; ++c) {
      int v = lb[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0 && v) atomicAdd(&h_before[c], v);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int r = 0, unaligned = 0;
    for (int c = 0; c < kClasses; ++c) {
      start[c] = r;
      base[c] = r + h_before[c] * (32 >> c);
      unaligned += h_all[c] * (32 >> c);
      r = (r + h_all[c] * (32 >> c) + 127) & ~127;
    }
    start[kClasses] = r;
    if (blockIdx.x == 0) {
      hdr[0] = r;
      for (int c = 0; c < kClasses; ++c) hdr[1 + c] = c + 1 < kClasses ? start[c + 1] : r;
      hdr[7] = total;
      hdr[8] = h_all[kClasses];
      hdr[9] = unaligned;
      hdr[10] = 0;                   // tile ticket and exit counter of the persistent group kernels (mlp_group.hip: g_draw_ticket)
      hdr[11] = 0;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0) {   // alignment rows at the end of every region
    for (int c = 0; c < kClasses; ++c) {
      const int e = start[c] + h_all[c] * (32 >> c);
      for (int r = e + tid; r < start[c + 1]; r += 256) { crow_p[r] = 0; crow_c[r] = -1; }
    }
  }
  const int i0 = blockIdx.x * 256, i = i0 + tid;
  const bool ok = i < total;
  int rows = 0;
  const int mask = ok ? parts_of(cnt[i], ns, smin, split, &rows) : 0;
  int rank[kClasses];
#pragma unroll
  for (int cc = 0; cc < kClasses; ++cc) {
    const unsigned long long mk = __ballot((mask >> cc) & 1);
    rank[cc] = __popcll(mk & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave][cc] = __popcll(mk);
  }
  __syncthreads();
#pragma unroll
  for (int cc = 0; cc < kClasses; ++cc) {
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w][cc];
    r0_s[tid][cc] = base[cc] + (before + rank[cc]) * (32 >> cc);
  }
  __syncthreads();
  // Row writes, wave-cooperative: `ns` lanes per centre (lane = slot t), 64 / ns centres of the wave's 64 per step, so that
  // a centre's index row is ONE coalesced read and every part of it one coalesced write (a thread per centre walking its
  // own <= 32 slots cost 102 us for SA1 on ray-cast scenes, whose balls are 0.3-0.9 full: strided reads, scattered writes).
  // The part that holds slot t (parts in descending size) is found from the highest bit in which t and `rows` differ — it
  // is set in `rows` (t < rows) and the bits above it agree, so t lies in the part of that size, which starts at those
  // common bits.
  __shared__ int rows_s[256], tag_s[256];
  {
    int tag = i;
    if (ok && cnt[i] <= 0) tag |= 0x40000000;      // empty ball: pooled value 0
    if (rows & (rows - 1)) tag |= 0x20000000;      // several parts: combine with an atomic max
    rows_s[tid] = ok ? rows : 0;
    tag_s[tid] = tag;
  }
  __syncthreads();
  // The rows of the pooled buffer that MULTI-PART centres max-combine into (integer atomic max) start at zero: cleared here,
  // one launch earlier than their first writer, instead of by a separate fill.  A single-part centre's row is written whole
  // by a plain store and needs no clearing (round 2 cleared every row: 166 MB of writes per 32-scene pass).
  if (zero_y) {
    const int w4 = width >> 2, nrow = total - i0 < 256 ? total - i0 : 256;
    for (int e = tid; e < nrow * w4; e += 256) {
      const int r = e / w4, c = e - r * w4;
      const int rw = rows_s[r];
      if (rw & (rw - 1)) *reinterpret_cast<float4 *>(zero_y + (size_t)(i0 + r) * ldy + col0 + 4 * c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const int per = 64 / ns > 0 ? 64 / ns : 1;       // centres per step (ns = 32: 2, 16: 4, ... ; ns > 64 does not occur)
  const int t = lane % ns, sub = lane / ns;
  for (int c0 = 0; c0 < 64; c0 += per) {
    const int lc = wave * 64 + c0 + sub;           // centre of this lane inside the block
    const int rws = sub < per ? rows_s[lc] : 0;
    if (t < rws) {
      const int ci = i0 + lc;
      const int pbit = 31 - __builtin_clz(rws ^ t), sz = 1 << pbit;
      const int row = r0_s[lc][5 - pbit] + (t - (rws & ~(2 * sz - 1)));
      crow_p[row] = (ci / m) * n + idx[(size_t)ci * ns + t];
      crow_c[row] = tag_s[lc];
    }
  }
}

}  // namespace

DET6D_API int det6d_compact_rows_capacity(int total_centres, int ns) {
  return (total_centres * ns + kClasses * 128 + 1023) & ~1023;   // a multiple of 8 row tiles: keeps the XCD-aware tile order
}

DET6D_API int det6d_compact_hdr_ints(int total_centres) { return 16 + (kClasses + 1) * (det6d_divup(total_centres, 256) + 1); }

int det6d_compact_check_group(int ns, int smin, int *split) {
  if (ns != 1 && ns != 2 && ns != 4 && ns != 8 && ns != 16 && ns != 32) return DET6D_EINVAL;
  if (smin < 1 || smin > ns || (smin & (smin - 1))) return DET6D_EINVAL;
  if (*split < 0 || (*split & (*split - 1)) || (*split && *split < smin)) return DET6D_EINVAL;
  if (*split > ns) *split = ns;
  return DET6D_OK;
}

int det6d_compact_split_tol(int split) {
  // experiment knob: DET6D_COMPACT_TOL=t keeps a centre in ONE power-of-two part when that wastes <= 1/t of its rows
  static const int tol = det6d_env_int("DET6D_COMPACT_TOL", 0);
  return tol > 0 ? split | (tol << 8) : split;
}

static int check_group(int ns, int smin, int *split, const int *cnt, const int *idx, const int *hdr, const int *crow_p,
                       const int *crow_c) {
  if (!cnt || !idx || !hdr || !crow_p || !crow_c) return DET6D_EINVAL;
  return det6d_compact_check_group(ns, smin, split);
}

static int launch_groups(int b, int n, int m, int ngroups, const PairArgs &pa_in, float *zero_y, int ldy, hipStream_t stream,
                         bool counted = false) {
  PairArgs pa = pa_in;
  pa.g[0].split = det6d_compact_split_tol(pa.g[0].split);
  pa.g[1].split = det6d_compact_split_tol(pa.g[1].split);
  const int total = b * m;
  const int nblk = det6d_divup(total, 256) > 0 ? det6d_divup(total, 256) : 1;
  // counted: the per-block part counts are in the tables already (det6d_ball_query_pair_grid_lists leaves them there)
  if (!counted) hipLaunchKernelGGL(compact_count_kernel, dim3(nblk, ngroups), dim3(256), 0, stream, total, pa);
  hipLaunchKernelGGL(compact_place_kernel, dim3(nblk, ngroups), dim3(256), 0, stream, total, n, m, nblk, zero_y, ldy, pa);
  return det6d_check_launch("det6d_compact_groups");
}

DET6D_API int det6d_compact_groups(int b, int n, int m, int ns, int smin, int split, const int *cnt, const int *idx,
                                   int *hdr, int *crow_p, int *crow_c, float *zero_y, int ldy, int col0, int width,
                                   det6d_stream_t stream) {
  if (b < 0 || n <= 0 || m <= 0) return DET6D_EINVAL;
  if (check_group(ns, smin, &split, cnt, idx, hdr, crow_p, crow_c)) return DET6D_EINVAL;
  if (zero_y && ((ldy | col0 | width) & 3 || ((uintptr_t)zero_y & 15) || width <= 0 || col0 + width > ldy)) return DET6D_EINVAL;
  PairArgs pa;
  pa.g[0] = GroupArgs{ns, smin, split, cnt, idx, hdr, crow_p, crow_c, col0, width};
  pa.g[1] = pa.g[0];
  return launch_groups(b, n, m, 1, pa, zero_y, ldy, (hipStream_t)stream);
}

// both radius groups of an SA layer (same centres, their own nsample / counts / indices / lists) in one pair of launches
static int groups_pair(int b, int n, int m, int smin, int split, int ns_a, const int *cnt_a, const int *idx_a,
                       int *hdr_a, int *crow_p_a, int *crow_c_a, int col0_a, int width_a, int ns_b,
                       const int *cnt_b, const int *idx_b, int *hdr_b, int *crow_p_b, int *crow_c_b, int col0_b,
                       int width_b, float *zero_y, int ldy, det6d_stream_t stream, bool counted) {
  if (b < 0 || n <= 0 || m <= 0) return DET6D_EINVAL;
  int sa = split, sb = split;
  const int smin_a = smin < ns_a ? smin : ns_a, smin_b = smin < ns_b ? smin : ns_b;
  if (sa && sa < smin_a) sa = smin_a;
  if (sb && sb < smin_b) sb = smin_b;
  if (check_group(ns_a, smin_a, &sa, cnt_a, idx_a, hdr_a, crow_p_a, crow_c_a)) return DET6D_EINVAL;
  if (check_group(ns_b, smin_b, &sb, cnt_b, idx_b, hdr_b, crow_p_b, crow_c_b)) return DET6D_EINVAL;
  if (zero_y && ((ldy | col0_a | width_a | col0_b | width_b) & 3 || ((uintptr_t)zero_y & 15) || width_a <= 0 || width_b <= 0 ||
                 col0_a + width_a > ldy || col0_b + width_b > ldy))
    return DET6D_EINVAL;
  PairArgs pa;
  pa.g[0] = GroupArgs{ns_a, smin_a, sa, cnt_a, idx_a, hdr_a, crow_p_a, crow_c_a, col0_a, width_a};
  pa.g[1] = GroupArgs{ns_b, smin_b, sb, cnt_b, idx_b, hdr_b, crow_p_b, crow_c_b, col0_b, width_b};
  return launch_groups(b, n, m, 2, pa, zero_y, ldy, (hipStream_t)stream, counted);
}

DET6D_API int det6d_compact_groups_pair(int b, int n, int m, int smin, int split, int ns_a, const int *cnt_a, const int *idx_a,
                                        int *hdr_a, int *crow_p_a, int *crow_c_a, int col0_a, int width_a, int ns_b,
                                        const int *cnt_b, const int *idx_b, int *hdr_b, int *crow_p_b, int *crow_c_b, int col0_b,
                                        int width_b, float *zero_y, int ldy, det6d_stream_t stream) {
  return groups_pair(b, n, m, smin, split, ns_a, cnt_a, idx_a, hdr_a, crow_p_a, crow_c_a, col0_a, width_a, ns_b, cnt_b, idx_b, hdr_b,
                     crow_p_b, crow_c_b, col0_b, width_b, zero_y, ldy, stream, false);
}

// the list builder behind det6d_ball_query_pair_grid_lists(smin, split, hdr_a, hdr_b), which has left the per-block part counts
// in the two hdr buffers: placement only.  m must be a multiple of 256 (the query's workgroups are the builder's blocks).
DET6D_API int det6d_compact_groups_pair_counted(int b, int n, int m, int smin, int split, int ns_a, const int *cnt_a,
                                                const int *idx_a, int *hdr_a, int *crow_p_a, int *crow_c_a, int col0_a, int width_a,
                                                int ns_b, const int *cnt_b, const int *idx_b, int *hdr_b, int *crow_p_b,
                                                int *crow_c_b, int col0_b, int width_b, float *zero_y, int ldy,
                                                det6d_stream_t stream) {
  if (m % 256 != 0) return DET6D_EINVAL;
  return groups_pair(b, n, m, smin, split, ns_a, cnt_a, idx_a, hdr_a, crow_p_a, crow_c_a, col0_a, width_a, ns_b, cnt_b, idx_b, hdr_b,
                     crow_p_b, crow_c_b, col0_b, width_b, zero_y, ldy, stream, true);
}
