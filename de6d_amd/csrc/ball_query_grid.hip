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
#include "compact_parts.h"

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

// PACKED (n <= 32768): two 16-bit counters per LDS word (a counter never exceeds n, so no carry crosses the halves): 33 KB of
// LDS instead of 66 — a workgroup of this kernel then fits on a CU beside the head's wide group kernel (99 KB); with 66 KB it
// had to wait for a CU without one, holding its hardware queue's dispatch slot meanwhile.
template <bool PACKED>
__global__ __launch_bounds__(kBuildThreads) void bq_grid_build_kernel(int n, float cell, const float *__restrict__ xyz,
                                                                      GridHeader *__restrict__ hdr,
                                                                      int *__restrict__ cell_start,
                                                                      float4 *__restrict__ sorted_pts) {
  __shared__ unsigned counts[PACKED ? kGridCells / 2 : kGridCells];
  auto cget = [&](int c) -> int { return PACKED ? (int)((counts[c >> 1] >> (16 * (c & 1))) & 0xffffu) : (int)counts[c]; };
  auto cset = [&](int c, int v) {          // (only ever called by the one thread that owns cells c and c ^ 1)
    if (PACKED) counts[c >> 1] = (counts[c >> 1] & ~(0xffffu << (16 * (c & 1)))) | ((unsigned)v << (16 * (c & 1)));
    else counts[c] = (unsigned)v;
  };
  auto cadd = [&](int c) -> int {          // counter of cell c += 1, returns the old value
    if (PACKED) return (int)((atomicAdd(&counts[c >> 1], 1u << (16 * (c & 1))) >> (16 * (c & 1))) & 0xffffu);
    return (int)atomicAdd(&counts[c], 1u);
  };
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
  for (int c = tid; c < (PACKED ? (ncells + 1) / 2 : ncells); c += kBuildThreads) counts[c] = 0u;
  __syncthreads();
  for (int k = tid; k < n; k += kBuildThreads) {
    const int cx = cell_coord(xyz[(size_t)k * 3 + 0], xmin, inv, nx);
    const int cy = cell_coord(xyz[(size_t)k * 3 + 1], ymin, inv, ny);
    cadd(cy * nx + cx);
  }
  __syncthreads();
  // exclusive scan of the counts of cells [0, ncells): each thread owns a contiguous chunk (an EVEN number of cells, so that
  // the two counters of a packed word belong to one thread)
  const int chunk = ((ncells + kBuildThreads - 1) / kBuildThreads + 1) & ~1;
  const int c0 = min(tid * chunk, ncells), c1 = min(c0 + chunk, ncells);
  int local = 0;
  for (int c = c0; c < c1; ++c) local += cget(c);
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
    const int cnt = cget(c);
    cell_start[c] = run;
    cset(c, run);                                  // becomes the scatter cursor (PACKED: run < n <= 32768 fits 16 bits)
    run += cnt;
  }
  if (tid == 0) cell_start[ncells] = n;
  __syncthreads();
  for (int k = tid; k < n; k += kBuildThreads) {
    const int cx = cell_coord(xyz[(size_t)k * 3 + 0], xmin, inv, nx);
    const int cy = cell_coord(xyz[(size_t)k * 3 + 1], ymin, inv, ny);
    const int pos = cadd(cy * nx + cx);
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

// ---- the query ---------------------------------------------------------------------------------------------------------
// The reference's contract is "the first nsample hits by ASCENDING POINT INDEX", but the grid hands out candidates in spatial
// order, so per shell a centre keeps the nsample smallest hit indices seen so far.
//
// Round 5: a workgroup owns 256 CONSECUTIVE centres of one scene, one per lane.  On FPS-sampled clouds a centre's 3 x 3 cells
// hold ~20 candidates; the round-2..4 kernel gave every centre a whole wave (three 64-lane sweeps with ~6 live lanes each,
// two ballots, two list rankings: ~275 vector instructions per centre, 49 M per 32-scene pass — the largest vector-ALU
// consumer beside the fp32 MFMAs it shares the issue port with).  Now
//   LIGHT centres (few candidates; the cut is chosen per wave, see the kernel): the lane walks its own candidates one by one and keeps its hits SORTED in a
//     per-lane LDS list (insertion from the back: candidates of a cell arrive nearly in index order);
//   HEAVY centres (dense parts of a real sweep: hundreds of candidates): the wave takes them one at a time, lane =
//     candidate, exactly as before (running selection in a kListCap-entry list, pruning threshold, ranked at the end);
//   OUTPUT: the wave writes its 64 centres' index rows as 16-byte stores out of the lane lists (cyclic padding as the
//     reference writes it: ball_query_gpu.cu:75-90,114-129), and — when the caller builds compact row lists from the result
//     (compact.hip) — the workgroup leaves the per-class part counts of its 256 centres in the list builder's table, which
//     saves that builder's counting launch.
constexpr int kListCap = 128;
constexpr int kLaneThreads = 256;
constexpr int kWalkShipped = 2;               // candidates a lane fetches per step of its walk
constexpr int kLightCap = 96;                // most candidates a lane walks by itself (experiments build: DET6D_BQ_LIGHT_CAP)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
}

// One wave, one centre (wave-uniform arguments): both shells' hit lists in la / lb (ascending, <= ns entries), counts returned.
// Hits are appended, unordered, through ballots; when a list would overflow it is cut back to its nsample smallest entries and
// the largest of them becomes a THRESHOLD — a later hit with a larger index can never be among the first nsample and is
// dropped before it is stored.  After the first cut the list holds >= nsample entries for good, so min(total, nsample) is
// known without counting the dropped hits.
__device__ __forceinline__ void wave_centre_scan(float qx, float qy, float qz, const int (&beg)[3], const int (&end)[3],
                                                 float rin2_a, float rout2_a, int ns_a, float rin2_b, float rout2_b, int ns_b,
                                                 const float4 *__restrict__ si, int *__restrict__ la, int *__restrict__ lb,
                                                 int *__restrict__ tw, int lane, int &cnt_a, int &cnt_b) {
  const unsigned long long below = (1ull << lane) - 1ull;
  int na = 0, nb = 0;                              // entries in the lists (wave-uniform)
  int thr_a = 0x7fffffff, thr_b = 0x7fffffff;      // only hits with a smaller point index can still be among the first nsample
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    for (int t0 = beg[r]; t0 < end[r]; t0 += 64) {
      const int t = t0 + lane;
      bool ha = false, hb = false;
      int k = 0;
      if (t < end[r]) {
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
  cnt_a = keep_smallest(la, tw, na, ns_a, lane);   // = min(total hits, ns)
  cnt_b = keep_smallest(lb, tw, nb, ns_b, lane);
}

// a lane's sorted list lives at L[i * kLaneThreads] (i = 0 .. ns - 1; the caller has added its thread index): every lane of a
// wave touches its own bank whatever its i.  `last` = the largest kept entry (-1: none yet), held in a register: the candidates
// of a cell row arrive nearly in index order, so the common hit is an APPEND — one LDS store, no LDS read on the walk's
// dependency chain; a hit that belongs inside the list shifts the larger entries up.
template <typename LT>
__device__ __forceinline__ void lane_insert(LT *__restrict__ L, int &cnt, int &last, int ns, int k) {
  if (k > last) {
    if (cnt < ns) { L[cnt * kLaneThreads] = (LT)k; ++cnt; last = k; }     // (full: not among the ns smallest)
    return;
  }
  const bool full = cnt == ns;
  int pos = full ? ns - 1 : cnt++;                  // the slot that opens at the end (full: the largest entry falls out)
  while (pos > 0) {
    const int v = (int)L[(pos - 1) * kLaneThreads];
    if (v < k) break;
    L[pos * kLaneThreads] = (LT)v;
    --pos;
  }
  L[pos * kLaneThreads] = (LT)k;
  if (full) last = (int)L[(ns - 1) * kLaneThreads];
}

struct QueryArgs {
  int n, m;
  float rin2_a, rout2_a;
  int ns_a;
  float rin2_b, rout2_b;
  int ns_b;
  const float *new_xyz;
  const GridHeader *hdr;
  const int *cell_start;
  const float4 *sorted_pts;
  int *cnt[2], *idx[2];
  CompactCountArgs count[2];
  int light_cap, fixed_cut;
};

// l mod c for 0 <= l < 4096, 1 <= c <= 64 (rc ~ 1 / c): the quotient estimate is exact or one short
__device__ __forceinline__ int small_mod(int l, int c, float rc) {
  const int q = (int)((float)l * rc);
  const int r = l - q * c;
  return r >= c ? r - c : r;
}

template <typename LT, bool PAD, int kWalk>
__global__ __launch_bounds__(kLaneThreads) void bq_grid_query_kernel(const QueryArgs qa) {
  extern __shared__ __align__(16) unsigned char lane_lists_raw[];     // (ns_a + ns_b) x 256 entries
  __shared__ int lists[kLaneThreads / 64][2][kListCap];
  __shared__ int tmp[kLaneThreads / 64][kMaxNs];
  __shared__ int cnts[2][kLaneThreads];
  __shared__ int acc[2][kCompactClasses + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bs = blockIdx.y, n = qa.n, m = qa.m, ns_a = qa.ns_a, ns_b = qa.ns_b;
  const GridHeader h = qa.hdr[bs];
  const int *cs = qa.cell_start + (size_t)bs * (kGridCells + 1);
  const float4 *si = qa.sorted_pts + (size_t)bs * n;
  LT *LA = reinterpret_cast<LT *>(lane_lists_raw) + tid, *LB = LA + (size_t)ns_a * kLaneThreads;
  const int ci = blockIdx.x * kLaneThreads + tid;
  const bool valid = ci < m;
  if (tid <= kCompactClasses) { acc[0][tid] = 0; acc[1][tid] = 0; }

  float qx = 0.f, qy = 0.f, qz = 0.f;
  int beg[3] = {0, 0, 0}, end[3] = {0, 0, 0};
  if (valid) {
    const float *q = qa.new_xyz + ((size_t)bs * m + ci) * 3;
    qx = q[0]; qy = q[1]; qz = q[2];
    const int cx = cell_coord(qx, h.ox, h.inv_cell, h.nx), cy = cell_coord(qy, h.oy, h.inv_cell, h.ny);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, h.nx - 1);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int y = cy - 1 + r;
      if (y >= 0 && y < h.ny) { beg[r] = cs[y * h.nx + x0]; end[r] = cs[y * h.nx + x1 + 1]; }   // x-contiguous cells: one range
    }
  }
  const int n0 = end[0] - beg[0], n01 = n0 + end[1] - beg[1], cand = n01 + end[2] - beg[2];
  // Light or heavy, decided per WAVE: a wave walks its light lanes in lockstep, so the walk costs the wave max(cand of its
  // light lanes) steps of ~40 vector instructions whatever the other lanes hold, and a heavy centre ~300 for the wave.  The
  // cut T that minimises  80 T + 300 #(cand > T)  over a few candidates (five ballots, scalar arithmetic) keeps a sparse wave
  // from following one dense centre through 200 steps, and lets a uniformly dense wave (65536-point scenes: ~70 candidates
  // per centre) walk instead of taking 64 turns.  (80 rather than 40 per step: a step is also a dependent L2 round trip, and
  // the launch ends with its longest walk — with 40 the first layer's query took 153 us on an idle chip, with a cut pinned at
  // 32 or 48 it takes 85; scripts/r05/gpu_t6.sh.)
  int cut = 0;
  {
    int best_cost = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      constexpr int kCuts[5] = {16, 32, 48, 64, 96};
      const int T = kCuts[t];
      if (T > qa.light_cap && t > 0) break;
      if (qa.fixed_cut > 0 && T != qa.fixed_cut) continue;      // (experiments build: DET6D_BQ_CUT pins the cut)
      const int cost = 80 * T + 300 * (int)__popcll(__ballot(cand > T));
      if (cost < best_cost) { best_cost = cost; cut = T; }
    }
  }
  const bool heavy = cand > cut;
  const int walk = heavy ? 0 : cand;
  int ca = 0, cb = 0, last_a = -1, last_b = -1;

  // ---- light centres: lane = centre, kWalk candidates per step (their records are fetched together: the walk is a chain of
  // L2 round trips, not of arithmetic)
  {
    const float rin2_a = qa.rin2_a, rout2_a = qa.rout2_a, rin2_b = qa.rin2_b, rout2_b = qa.rout2_b;
    for (int j0 = 0; __ballot(j0 < walk) != 0ull; j0 += kWalk) {
      float4 c[kWalk];
#pragma unroll
      for (int u = 0; u < kWalk; ++u) {
        const int j = j0 + u;
        c[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < walk) c[u] = si[j < n0 ? beg[0] + j : (j < n01 ? beg[1] + (j - n0) : beg[2] + (j - n01))];
      }
#pragma unroll
      for (int u = 0; u < kWalk; ++u) {
        if (j0 + u < walk) {
          const int k = __float_as_int(c[u].w);
          const float d2 = d6_sqdist(qx - c[u].x, qy - c[u].y, qz - c[u].z);
          if (d2 >= rin2_a && d2 < rout2_a) lane_insert<LT>(LA, ca, last_a, ns_a, k);
          if (d2 >= rin2_b && d2 < rout2_b) lane_insert<LT>(LB, cb, last_b, ns_b, k);
        }
      }
    }
  }

  // ---- heavy centres of this wave, one at a time: lane = candidate
  {
    int *la = lists[wave][0], *lb = lists[wave][1], *tw = tmp[wave];
    unsigned long long hm = __ballot(heavy);
    while (hm != 0ull) {
      const int src = __builtin_ctzll(hm);
      hm &= hm - 1ull;
      int hb[3], he[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) { hb[r] = __builtin_amdgcn_readlane(beg[r], src); he[r] = __builtin_amdgcn_readlane(end[r], src); }
      int wa, wb;
      wave_centre_scan(d6_readlane_f(qx, src), d6_readlane_f(qy, src), d6_readlane_f(qz, src), hb, he, qa.rin2_a, qa.rout2_a, ns_a,
                       qa.rin2_b, qa.rout2_b, ns_b, si, la, lb, tw, lane, wa, wb);
      LT *ca_col = reinterpret_cast<LT *>(lane_lists_raw) + (wave * 64 + src);
      if (lane < wa) ca_col[lane * kLaneThreads] = (LT)la[lane];                                   // ns <= 64: one entry per lane
      if (lane < wb) ca_col[(size_t)(ns_a + lane) * kLaneThreads] = (LT)lb[lane];
      if (lane == src) { ca = wa; cb = wb; }
      wave_lds_sync();                             // the wave lists are rewritten by the next heavy centre
    }
  }
  cnts[0][tid] = ca;
  cnts[1][tid] = cb;
  if (valid) { qa.cnt[0][(size_t)bs * m + ci] = ca; qa.cnt[1][(size_t)bs * m + ci] = cb; }
  wave_lds_sync();

  // ---- output: this wave's 64 centres, index rows out of the lane lists (row l of a ball with cnt hits = its hit l mod cnt)
#pragma unroll
  for (int sh = 0; sh < 2; ++sh) {
    const int ns = sh == 0 ? ns_a : ns_b;
    const LT *Lw = reinterpret_cast<const LT *>(lane_lists_raw) + (sh == 0 ? 0 : (size_t)ns_a * kLaneThreads) + wave * 64;
    int *out = qa.idx[sh] + ((size_t)bs * m + blockIdx.x * kLaneThreads + wave * 64) * ns;
    const int live = min(64, m - ((int)blockIdx.x * kLaneThreads + wave * 64)); // centres of this wave that exist (may be <= 0; int, not
                                                                                // blockIdx's unsigned: a negative count must stay negative)
    if ((ns & 3) == 0) {
      const int gper = ns >> 2;                    // 16-byte groups per centre
      const float rg = __builtin_amdgcn_rcpf((float)gper);
      for (int it = 0; it < gper; ++it) {
        const int g = it * 64 + lane;
        int c = (int)((float)g * rg);              // g / gper, exact or one short (g < 1024)
        if (g - c * gper >= gper) ++c;
        const int l0 = (g - c * gper) << 2;
        const int cnt = cnts[sh][wave * 64 + c];
        const int cm = max(cnt, 1);
        const float rc = __builtin_amdgcn_rcpf((float)cm);
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = (int)Lw[small_mod(l0 + u, cm, rc) * kLaneThreads + c];
          v[u] = cnt > 0 ? e : 0;
        }
        // without padding (the caller reads compact rows only: compact.hip walks <= the next power of two >= max(cnt, 4) slots)
        const int need = cnt <= 4 ? 4 : (1 << (32 - __builtin_clz(cnt - 1)));
        if (c < live && (PAD || l0 < need)) *reinterpret_cast<int4 *>(out + (size_t)c * ns + l0) = make_int4(v[0], v[1], v[2], v[3]);
      }
    } else {                                        // any other nsample: one element per lane and step
      const float rn = __builtin_amdgcn_rcpf((float)ns);
      for (int it = 0; it < ns; ++it) {
        const int e = it * 64 + lane;
        int c = (int)((float)e * rn);
        if (e - c * ns >= ns) ++c;
        const int l = e - c * ns;
        const int cnt = cnts[sh][wave * 64 + c];
        const int cm = max(cnt, 1);
        const int v = (int)Lw[small_mod(l, cm, __builtin_amdgcn_rcpf((float)cm)) * kLaneThreads + c];
        if (c < live) out[(size_t)c * ns + l] = cnt > 0 ? v : 0;
      }
    }
  }

  // ---- per-class part counts of the 256 centres for the compact list builder (compact.hip: compact_place_kernel)
  if (qa.count[0].table != nullptr) {              // (uniform; the host passes tables only when m is a multiple of 256)
    __syncthreads();                               // acc cleared
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
      const CompactCountArgs &ct = qa.count[sh];
      const int k = sh == 0 ? ca : cb;
      int rows = 0;
      const int mask = valid ? d6_compact_parts_of(k, ct.ns, ct.smin, ct.split, &rows) : 0;
      int real = valid ? (k < ct.ns ? k : ct.ns) : 0;
#pragma unroll
      for (int cc = 0; cc < kCompactClasses; ++cc) {
        const int c = __popcll(__ballot((mask >> cc) & 1));
        if (lane == 0 && c) atomicAdd(&acc[sh][cc], c);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) real += __shfl_xor(real, o);
      if (lane == 0 && real) atomicAdd(&acc[sh][kCompactClasses], real);
    }
    __syncthreads();
    const int bk = bs * gridDim.x + blockIdx.x;
    if (tid <= kCompactClasses) {
      qa.count[0].table[bk * (kCompactClasses + 1) + tid] = acc[0][tid];
      qa.count[1].table[bk * (kCompactClasses + 1) + tid] = acc[1][tid];
    }
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

static int launch_grid_query(int b, int n, int m, float rin_a, float rout_a, int ns_a, float rin_b, float rout_b, int ns_b,
                             const float *new_xyz, const float *xyz, void *workspace, int *cnt_a, int *idx_a, int *cnt_b,
                             int *idx_b, bool pad, const CompactCountArgs *count, hipStream_t s, const char *what) {
  // index rows leave as 16-byte stores when nsample is a multiple of 4: the idx buffers must then be 16-byte aligned
  // (det6d_ops.h says so; checked BEFORE anything is queued)
  if (((ns_a & 3) == 0 && ((uintptr_t)idx_a & 15)) || ((ns_b & 3) == 0 && ((uintptr_t)idx_b & 15))) return DET6D_EINVAL;
  // workspace layout: headers | cell_start | sorted (x, y, z, index) records
  char *ws = (char *)workspace;
  GridHeader *hdr = (GridHeader *)ws;
  size_t off = ((size_t)b * sizeof(GridHeader) + 63) / 64 * 64;
  int *cell_start = (int *)(ws + off);
  off += ((size_t)b * (kGridCells + 1) * 4 + 63) / 64 * 64;
  float4 *sorted_pts = (float4 *)(ws + off);
  const float rmax = rout_a > rout_b ? rout_a : rout_b;
  // (the build kernel's PACKED 16-bit cell counters hold scenes of <= 32768 points; the query's 16-bit per-lane hit lists
  // hold point indices < 65536: two different limits, `narrow` below)
  if (n <= 32768) hipLaunchKernelGGL(bq_grid_build_kernel<true>, dim3(b), dim3(kBuildThreads), 0, s, n, rmax, xyz, hdr, cell_start, sorted_pts);
  else hipLaunchKernelGGL(bq_grid_build_kernel<false>, dim3(b), dim3(kBuildThreads), 0, s, n, rmax, xyz, hdr, cell_start, sorted_pts);
  QueryArgs qa;
  qa.n = n; qa.m = m;
  qa.rin2_a = rin_a * rin_a; qa.rout2_a = rout_a * rout_a; qa.ns_a = ns_a;
  qa.rin2_b = rin_b * rin_b; qa.rout2_b = rout_b * rout_b; qa.ns_b = ns_b;
  qa.new_xyz = new_xyz; qa.hdr = hdr; qa.cell_start = cell_start; qa.sorted_pts = sorted_pts;
  qa.cnt[0] = cnt_a; qa.cnt[1] = cnt_b; qa.idx[0] = idx_a; qa.idx[1] = idx_b;
  for (int g = 0; g < 2; ++g) qa.count[g] = count ? count[g] : CompactCountArgs{0, 0, 0, nullptr};
  static const int light_cap = det6d_env_int("DET6D_BQ_LIGHT_CAP", kLightCap);
  qa.light_cap = light_cap < 0 ? 0 : light_cap;
  static const int fixed_cut = det6d_env_int("DET6D_BQ_CUT", 0);
  qa.fixed_cut = fixed_cut;
  const bool narrow = n <= 65536;                  // point indices fit 16 bits: half the LDS per workgroup
  const size_t lds = (size_t)(ns_a + ns_b) * kLaneThreads * (narrow ? 2 : 4);
  const dim3 grid(det6d_divup(m, kLaneThreads), b), block(kLaneThreads);
#define D6_BQ_LAUNCH_W(LT, PAD, W)                                                             \
  do {                                                                                         \
    DET6D_MAX_DYNAMIC_LDS((bq_grid_query_kernel<LT, PAD, W>), 2 * kMaxNs * kLaneThreads * 4);   \
    hipLaunchKernelGGL((bq_grid_query_kernel<LT, PAD, W>), grid, block, lds, s, qa);            \
  } while (0)
#ifdef DET6D_EXPERIMENTS      // candidates fetched per walk step: 1 / 2 / 4 (DET6D_BQ_WALK) for A/B runs
  static const int walk = det6d_env_int("DET6D_BQ_WALK", kWalkShipped);
#define D6_BQ_LAUNCH(LT, PAD)                                                                  \
  do {                                                                                         \
    if (walk == 1) D6_BQ_LAUNCH_W(LT, PAD, 1); else if (walk == 2) D6_BQ_LAUNCH_W(LT, PAD, 2); else D6_BQ_LAUNCH_W(LT, PAD, 4); \
  } while (0)
#else
#define D6_BQ_LAUNCH(LT, PAD) D6_BQ_LAUNCH_W(LT, PAD, kWalkShipped)
#endif
  if (narrow) { if (pad) D6_BQ_LAUNCH(unsigned short, true); else D6_BQ_LAUNCH(unsigned short, false); }
  else { if (pad) D6_BQ_LAUNCH(int, true); else D6_BQ_LAUNCH(int, false); }
#undef D6_BQ_LAUNCH_W
#undef D6_BQ_LAUNCH
  return det6d_check_launch(what);
}

DET6D_API int det6d_ball_query_pair_grid(int b, int n, int m, float rin_a, float rout_a, int ns_a, float rin_b,
                                         float rout_b, int ns_b, const float *new_xyz, const float *xyz,
                                         void *workspace, int *cnt_a, int *idx_a, int *cnt_b, int *idx_b,
                                         det6d_stream_t stream) {
  if (b < 0 || m < 0 || !det6d_ball_query_grid_supported(n, ns_a, ns_b) || !new_xyz || !xyz ||
      !workspace || ((uintptr_t)workspace & 15) || !cnt_a || !idx_a || !cnt_b || !idx_b)
    return DET6D_EINVAL;
  if (b == 0 || m == 0) return DET6D_OK;
  return launch_grid_query(b, n, m, rin_a, rout_a, ns_a, rin_b, rout_b, ns_b, new_xyz, xyz, workspace, cnt_a, idx_a, cnt_b,
                           idx_b, true, nullptr, (hipStream_t)stream, "det6d_ball_query_pair_grid");
}

// Engine form (fused section of the header): the same query feeding det6d_compact_groups_pair_counted.  Index rows are written
// only as far as the compact list builder reads them (slots below the next power of two >= max(cnt, 4), cyclic padding
// included), and the builder's per-block part counts are left in hdr_a / hdr_b (m must be a multiple of 256).
DET6D_API int det6d_ball_query_pair_grid_lists(int b, int n, int m, float rin_a, float rout_a, int ns_a, float rin_b,
                                               float rout_b, int ns_b, const float *new_xyz, const float *xyz,
                                               void *workspace, int *cnt_a, int *idx_a, int *cnt_b, int *idx_b,
                                               int smin, int split, int *hdr_a, int *hdr_b, det6d_stream_t stream) {
  if (b < 0 || m < 0 || !det6d_ball_query_grid_supported(n, ns_a, ns_b) || !new_xyz || !xyz ||
      !workspace || ((uintptr_t)workspace & 15) || !cnt_a || !idx_a || !cnt_b || !idx_b || !hdr_a || !hdr_b)
    return DET6D_EINVAL;
  if (m % kLaneThreads != 0 || smin < 1 || smin > 4 || (smin & (smin - 1))) return DET6D_EINVAL;
  if (b == 0 || m == 0) return DET6D_OK;
  int sa = split, sb = split;
  const int smin_a = smin < ns_a ? smin : ns_a, smin_b = smin < ns_b ? smin : ns_b;
  if (sa && sa < smin_a) sa = smin_a;              // (as det6d_compact_groups_pair does)
  if (sb && sb < smin_b) sb = smin_b;
  if (det6d_compact_check_group(ns_a, smin_a, &sa) || det6d_compact_check_group(ns_b, smin_b, &sb)) return DET6D_EINVAL;
  const CompactCountArgs count[2] = {{ns_a, smin_a, det6d_compact_split_tol(sa), hdr_a + 16},
                                     {ns_b, smin_b, det6d_compact_split_tol(sb), hdr_b + 16}};
  return launch_grid_query(b, n, m, rin_a, rout_a, ns_a, rin_b, rout_b, ns_b, new_xyz, xyz, workspace, cnt_a, idx_a, cnt_b,
                           idx_b, false, count, (hipStream_t)stream, "det6d_ball_query_pair_grid_lists");
}
