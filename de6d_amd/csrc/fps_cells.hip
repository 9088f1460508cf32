// fps_cells.hip — the pre-pass of the 16384-point farthest point samplers (D-FPS) for gfx950, their launcher, and (experiments
// build only) the one-pick wave-skip sampler that shipped in rounds 2-3.
//
// Same result, bit for bit, as farthest_point_sampling_kernel
// (core/pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:101-222) including its tie order, but a pick no longer
// touches every point.
//
// Observation.  All min-distances satisfy temp[k] <= M, where M is the current maximum of a region.  A point can only change
// in this round if d(k, s) < temp[k] <= M.  Points are pre-sorted into a 4 x 4 k-d grid of equal counts (cell_sort_kernel),
// one cell per wave; a wave whose bounding box is at least sqrt(M) away from the new sample s cannot change, and what it
// published stays valid.  The box test is exact in floating point: subtraction, multiplication and fma are monotone, so
// lb = fma(gz,gz, fma(gx,gx, gy*gy))  with per-axis gaps g <= |x_k - s| is a true lower bound of the distance the kernel
// would compute for every point of the cell.  A new sample reaches 1.3 of the 16 boxes on average.
// Ties (exactly equal maxima: duplicated points) are resolved on a slow path with the reference's order:
// minimise (bitrev(k mod S), k).
//
// The sampling kernel is fps_seq.hip's multi-pick form (several picks per barrier round from published top-4 lists).  The
// wave-skip kernel below (one pick per round: rescan -> LDS slot -> barrier -> block arg-max) is compiled into the experiments
// build only (DET6D_FPS_SEQ=0): same-library A/B in the pipeline, round 4: 12.40-12.43 k scenes/s against 12.86-12.95 k
// (benchmark scenes), 5.63 against 5.73 k (ray-cast); one frame on an idle chip 4.7 against 3.7 ms.
#include "common.h"
#include <stdlib.h>

namespace {

// exclusive prefix sum over the 1024 threads of a workgroup: wave scan (six shuffles), the sixteen wave totals through LDS
// (`wtot`: 16 words; free again when the call returns).  (Rounds 2-4 used hipcub::BlockScan here.)
__device__ __forceinline__ unsigned block_exclusive_sum_1024(unsigned v, unsigned *__restrict__ wtot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = (unsigned)__shfl_up((int)incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  unsigned base = 0u;
  for (int w = 0; w < wave; ++w) base += wtot[w];
  __syncthreads();
  return base + incl - v;
}

__device__ __forceinline__ unsigned bitrev_bits(unsigned v, int bits) {
  return bits == 0 ? 0u : (__builtin_bitreverse32(v) >> (32 - bits));
}

// order key of point k under the reference's tie rule: smaller key wins
__device__ __forceinline__ unsigned tie_key(int k, int log2s) {
  return (bitrev_bits((unsigned)k & ((1u << log2s) - 1u), log2s) << (32 - log2s)) | ((unsigned)k >> log2s);
}

// lane holding the smallest key among the lanes of `cand` (slow path, ties only)
__device__ __forceinline__ int min_key_lane(unsigned long long cand, int k, int log2s) {
  const int lane = threadIdx.x & 63;
  const bool mine = (cand >> lane) & 1ull;
  unsigned key = mine ? tie_key(k, log2s) : 0xFFFFFFFFu;
  unsigned m = key;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)m, off);
    m = o < m ? o : m;
  }
  return __builtin_ctzll(__ballot(mine && key == m));
}

// ------------------------------------------------------------------------------------------------
// Pre-pass: spatial order of a scene.  perm[b, p] = original index of the point at sorted position p.
// Any permutation is CORRECT for the samplers below; the order only makes their regions compact.
// ------------------------------------------------------------------------------------------------
// src != nullptr (the cooperative 32768 / 65536-point sampler, fps_coop.hip): workgroup g orders the N points src[g N ..] of
// scene g / parts (one spatial PART of that scene, split off by coop_split_kernel) instead of a whole N-point scene.
template <int IPT>
__global__ __launch_bounds__(1024) void cell_sort_kernel(int n, int log2s, long long xyz_bstride, const float *__restrict__ xyz,
                                                         int *__restrict__ perm, const int *__restrict__ src = nullptr,
                                                         int parts = 1) {
  // A 4 x 4 k-d grid of equal counts: order by x, cut into 4 strips, order every strip by y, cut into 4: region g = sorted
  // positions [g n/16, (g+1) n/16).  A wave of the samplers owns one region, and its bounding box is a tight rectangle: a new
  // sample lands in (or within reach of) 1.3 of the 16 boxes on average.  A Morton curve cut into 16 equal runs gives 2.65: a
  // run that crosses a quadrant boundary of the curve has a box that spans both quadrants.
  // Two COUNTING sorts in LDS (histogram -> exclusive scan -> one atomic per point for its position): 1024 x bins decide the
  // strip, n / 16 y bins per strip the position inside it.  The order inside a bin is whatever the atomics make it — any
  // permutation is correct, and exact duplicates (same bin) stay within a bin's few points of each other, which is what lets
  // sq_hide_lane_duplicates find most of them in one lane.  (Rounds 2-4 ran two block radix sorts here: 124 us for 16384 points.)
  // LDS: 48 KB for 16384 points (4096 bins + 16-bit positions), so that a workgroup of this kernel fits on a CU beside the
  // head's wide group kernel (99 KB) — rounds 2-4 held 32-bit positions and one y bin per point: 131 KB, which no CU running
  // that kernel could take, and a pre-pass that waits for a CU keeps its hardware queue's dispatch slot while it waits.
  constexpr int N = 1024 * IPT;
  constexpr int BY = N / 16;                   // y bins per strip: four points per bin on average (8 cm of an 80 m scene)
  constexpr int QB = IPT / 4;                  // y bins per thread in the scan (4 strips x BY = N / 4 bins = 1024 QB)
  static_assert(IPT % 4 == 0 && 4 * BY >= 1024, "bins[] serves the 1024 x bins too");
  __shared__ unsigned scan_tmp[16];
  __shared__ unsigned bins[4 * BY];            // histogram, then running offsets (x pass: the first 1024)
  __shared__ unsigned short sorted[N];
  __shared__ float red[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  xyz += (size_t)(blockIdx.x / parts) * xyz_bstride;
  perm += (size_t)blockIdx.x * n;
  if (src) src += (size_t)blockIdx.x * n;
  float x[IPT], y[IPT];
  float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int k = src ? src[i * 1024 + tid] : i * 1024 + tid;   // (coalesced: the point a thread holds does not matter here)
    x[i] = xyz[(size_t)k * 3 + 0];
    y[i] = xyz[(size_t)k * 3 + 1];
    if (x[i] == x[i] && fabsf(x[i]) < 1e30f) { xmin = fminf(xmin, x[i]); xmax = fmaxf(xmax, x[i]); }
    if (y[i] == y[i] && fabsf(y[i]) < 1e30f) { ymin = fminf(ymin, y[i]); ymax = fmaxf(ymax, y[i]); }
  }
  xmin = d6_wave_min(xmin); xmax = d6_wave_max(xmax); ymin = d6_wave_min(ymin); ymax = d6_wave_max(ymax);
  if (lane == 0) { red[0][wave] = xmin; red[1][wave] = xmax; red[2][wave] = ymin; red[3][wave] = ymax; }
  bins[tid] = 0u;
  __syncthreads();
  xmin = red[0][0]; xmax = red[1][0]; ymin = red[2][0]; ymax = red[3][0];
  for (int w = 1; w < 16; ++w) {
    xmin = fminf(xmin, red[0][w]); xmax = fmaxf(xmax, red[1][w]);
    ymin = fminf(ymin, red[2][w]); ymax = fmaxf(ymax, red[3][w]);
  }
  const float sx = xmax > xmin ? 1023.0f / (xmax - xmin) : 0.f;
  const float sy = ymax > ymin ? (float)(BY - 1) / (ymax - ymin) : 0.f;
  // ---- x: strip of every point
  unsigned bin[IPT];
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    float fx = (x[i] - xmin) * sx;
    fx = fx == fx ? fminf(fmaxf(fx, 0.f), 1023.f) : 0.f;
    bin[i] = (unsigned)fx;
    atomicAdd(&bins[bin[i]], 1u);
  }
  __syncthreads();
  {
    const unsigned off = block_exclusive_sum_1024(bins[tid], scan_tmp);     // (returns behind a barrier: every bins[tid] read)
    bins[tid] = off;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const unsigned pos = atomicAdd(&bins[bin[i]], 1u);     // 0 .. N-1, each once: exactly N / 4 points per strip
    float fy = (y[i] - ymin) * sy;
    fy = fy == fy ? fminf(fmaxf(fy, 0.f), (float)(BY - 1)) : 0.f;
    bin[i] = (pos / (unsigned)(N / 4)) * (unsigned)BY + (unsigned)fy;
  }
  __syncthreads();
  // ---- y inside the strips: 4 BY bins (strip-major), thread t scans bins t QB .. t QB + QB - 1
#pragma unroll
  for (int i = 0; i < QB; ++i) bins[i * 1024 + tid] = 0u;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < IPT; ++i) atomicAdd(&bins[bin[i]], 1u);
  __syncthreads();
  {
    unsigned cnt[QB], total = 0u, base;
#pragma unroll
    for (int i = 0; i < QB; ++i) { cnt[i] = bins[tid * QB + i]; total += cnt[i]; }
    base = block_exclusive_sum_1024(total, scan_tmp);
#pragma unroll
    for (int i = 0; i < QB; ++i) { bins[tid * QB + i] = base; base += cnt[i]; }   // (every thread rewrites only its own bins)
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < IPT; ++i) sorted[atomicAdd(&bins[bin[i]], 1u)] = (unsigned short)(i * 1024 + tid);
  __syncthreads();
  // The IPT consecutive positions of this thread are exactly the slots of ONE lane of the samplers: order them by the
  // reference's tie key here (the strict '>' of a lane's scan then keeps the right point among equal values), in registers.
  unsigned key[IPT];
  int val[IPT];
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    val[i] = sorted[tid * IPT + i];
    if (src) val[i] = src[val[i]];             // position inside the part -> point of the scene
    key[i] = tie_key(val[i], log2s);
  }
#pragma unroll
  for (int i = 1; i < IPT; ++i) {
#pragma unroll
    for (int j = i; j > 0; --j) {
      const bool sw = key[j - 1] > key[j];
      const unsigned ka = sw ? key[j] : key[j - 1], kb = sw ? key[j - 1] : key[j];
      const int va = sw ? val[j] : val[j - 1], vb = sw ? val[j - 1] : val[j];
      key[j - 1] = ka; key[j] = kb; val[j - 1] = va; val[j] = vb;
    }
  }
#pragma unroll
  for (int i = 0; i < IPT; ++i) perm[tid * IPT + i] = val[i];
}

#ifdef DET6D_EXPERIMENTS      // ---- the one-pick wave-skip sampler: experiments build only (DET6D_FPS_SEQ=0)
// ------------------------------------------------------------------------------------------------
// Wave-skip sampler: the fat-thread kernel of fps.hip on the k-d sorted points, with ONE bounding box per
// wave.  A wave whose box is at least sqrt(its current maximum) away from the new sample cannot change
// and keeps its cached arg-max; only the waves near the sample rescan their 32 points per lane.  Same
// register footprint and round latency as the fat kernel, but ~1/5 of its vector-ALU instructions once
// the first few hundred samples are placed — and on gfx950 every VALU instruction of a co-resident kernel
// is time taken from the fp32 MFMAs of the GEMM waves on that SIMD.
// Exactness: (1) the box test is a floating-point lower bound of the distance the scan would compute
// (monotone ops, see the top of this file); (2) ties: the pre-pass orders the points of a lane by the
// reference's tie key, so the strict `>` of the scan keeps the right one inside a lane; ties between
// lanes / waves take the explicit min-key slow path.
// ------------------------------------------------------------------------------------------------
template <int LO, int HI, int N>
__device__ __forceinline__ void skip_pick(int ws, int wl, const float (&px)[N], const float (&py)[N],
                                          const float (&pz)[N], float &sx, float &sy, float &sz) {
  if constexpr (HI - LO == 1) {
    sx = d6_readlane_f(px[LO], wl);
    sy = d6_readlane_f(py[LO], wl);
    sz = d6_readlane_f(pz[LO], wl);
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (ws < MID) skip_pick<LO, MID>(ws, wl, px, py, pz, sx, sy, sz);
    else skip_pick<MID, HI>(ws, wl, px, py, pz, sx, sy, sz);
  }
}

typedef float f32x2s __attribute__((ext_vector_type(2)));

// G groups per wave: group g of a wave = slots g*SG .. (g+1)*SG-1 of its 64 lanes = 64*SG consecutive sorted
// points with their own bounding box and cached arg-max (G = 1: one box per wave).
// timing experiments only (DET6D_FPS_DBG=7): wall clock (100 MHz) at the first and after the last round of every scene
__device__ unsigned long long d6_fps_clock[2 * 64];

// PSHIFT (timing experiments only, DET6D_FPS_THIN): 2^PSHIFT workgroups per scene, each sampling ITS share of the sorted
// scene on its own (no exchange: WRONG picks) — the footprint of a sampler spread thinly over several CUs
template <int NW, int SLOTS, int G, int PSHIFT = 0>
__global__ __launch_bounds__(64 * NW) void fps_skip_kernel(int n, int m, int log2s, long long xyz_bstride,
                                                           long long idx_bstride, int idx_add, int dbg,
                                                           const float *__restrict__ xyz,
                                                           const int *__restrict__ perm, int *__restrict__ idxs) {
  constexpr int SG = SLOTS / G, HG = SG / 2;
  static_assert(SG * G == SLOTS && SG % 2 == 0, "group layout");
  __shared__ float4 slot_v[2][NW];   // (max, x, y, z) of every wave, double buffered by round parity
  __shared__ int slot_k[2][NW];
  __shared__ unsigned short korig[64 * NW * SLOTS];   // sorted position -> original index (n <= 65536)
  const int h = threadIdx.x, lane = h & 63, wave = h >> 6;
  if (D6_DBG_IS(9)) d6_sampler_priority();   // raised wave priority measured 1 % SLOWER in the pipeline (9668 vs 9750 scenes/s): DET6D_FPS_DBG=9 turns it on
  const int scene = blockIdx.x >> PSHIFT, part = blockIdx.x & ((1 << PSHIFT) - 1);
  xyz += (size_t)scene * xyz_bstride;
  perm += (size_t)scene * n + (size_t)part * (64 * NW * SLOTS);
  idxs += (size_t)scene * idx_bstride;

  float px[SLOTS], py[SLOTS], pz[SLOTS], pt[SLOTS];
  float lox[G], loy[G], loz[G], hix[G], hiy[G], hiz[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float ax = 3.0e38f, ay = 3.0e38f, az = 3.0e38f, bx = -3.0e38f, by = -3.0e38f, bz = -3.0e38f;
#pragma unroll
    for (int j = 0; j < SG; ++j) {
      const int s = g * SG + j;
      const int pos = ((wave * G + g) * 64 + lane) * SG + j;
      const int k = perm[pos];
      korig[pos] = (unsigned short)k;
      px[s] = xyz[(size_t)k * 3 + 0];
      py[s] = xyz[(size_t)k * 3 + 1];
      pz[s] = xyz[(size_t)k * 3 + 2];
      asm volatile("" : "+v"(px[s]), "+v"(py[s]), "+v"(pz[s]));
      pt[s] = 1e10f;
      ax = d6_vmin(ax, px[s]); bx = d6_vmax(bx, px[s]);
      ay = d6_vmin(ay, py[s]); by = d6_vmax(by, py[s]);
      az = d6_vmin(az, pz[s]); bz = d6_vmax(bz, pz[s]);
    }
    lox[g] = d6_wave_min(ax); hix[g] = d6_wave_max(bx);   // the group's bounding box (uniform)
    loy[g] = d6_wave_min(ay); hiy[g] = d6_wave_max(by);
    loz[g] = d6_wave_min(az); hiz[g] = d6_wave_max(bz);
  }
  __syncthreads();

  float cx = xyz[0], cy = xyz[1], cz = xyz[2];
  if (h == 0 && part == 0) idxs[0] = idx_add;
  if (D6_DBG_IS(7) && h == 0 && blockIdx.x < 64) d6_fps_clock[2 * blockIdx.x] = wall_clock64();
  // cached arg-max of every group of this wave (uniform)
  float cg_val[G], cg_x[G], cg_y[G], cg_z[G];
  int cg_k[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { cg_val[g] = __builtin_inff(); cg_x[g] = cg_y[g] = cg_z[g] = 0.f; cg_k[g] = 0; }

  for (int r = 1; r < m; ++r) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      // 1. can any point of this group change?  (lb <= every distance the scan would compute; cg_val >= every pt)
      const float gx = fmaxf(0.f, fmaxf(lox[g] - cx, cx - hix[g]));
      const float gy = fmaxf(0.f, fmaxf(loy[g] - cy, cy - hiy[g]));
      const float gz = fmaxf(0.f, fmaxf(loz[g] - cz, cz - hiz[g]));
      const float lb = d6_sqdist(gx, gy, gz);
      if (!(lb >= cg_val[g]) && !(D6_DBG_IS(3) && r > 1)) {   // wave-uniform branch (dbg 3: fixed per-round cost only)
        float best = -1.0f;
        int bs = 0;
        const f32x2s c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
        for (int q = 0; q < HG; ++q) {
          const int s0 = g * SG + 2 * q;
          const f32x2s dx = f32x2s{px[s0], px[s0 + 1]} - c2x;
          const f32x2s dy = f32x2s{py[s0], py[s0 + 1]} - c2y;
          const f32x2s dz = f32x2s{pz[s0], pz[s0 + 1]} - c2z;
          f32x2s d = dy * dy;
          d = __builtin_elementwise_fma(dx, dx, d);
          d = __builtin_elementwise_fma(dz, dz, d);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int s = s0 + e;
            const float t = d6_vmin(d[e], pt[s]);
            pt[s] = t;
            const bool up = t > best;
            bs = up ? 2 * q + e : bs;      // slot within the group
            best = up ? t : best;
          }
        }
        const float wmax = d6_wave_max(best);
        const unsigned long long tie = __ballot(best == wmax);
        int wl = __builtin_ctzll(tie);
        const int gbase = (wave * G + g) * 64;
        if (__popcll(tie) != 1) wl = min_key_lane(tie, (int)korig[(gbase + lane) * SG + bs], log2s);
        const int ws = d6_readlane_i(bs, wl);
        cg_val[g] = wmax;
        cg_k[g] = (int)korig[(gbase + wl) * SG + ws];
        float sx, sy, sz;
        if (g == 0) skip_pick<0, SG>(ws, wl, px, py, pz, sx, sy, sz);
        if (G > 1 && g == 1) skip_pick<(G > 1 ? SG : 0), (G > 1 ? 2 * SG : SG)>(ws + SG, wl, px, py, pz, sx, sy, sz);
        if (G > 2 && g == 2) skip_pick<(G > 2 ? 2 * SG : 0), (G > 2 ? 3 * SG : SG)>(ws + 2 * SG, wl, px, py, pz, sx, sy, sz);
        if (G > 3 && g == 3) skip_pick<(G > 3 ? 3 * SG : 0), (G > 3 ? 4 * SG : SG)>(ws + 3 * SG, wl, px, py, pz, sx, sy, sz);
        cg_x[g] = sx; cg_y[g] = sy; cg_z[g] = sz;
      }
    }
    // 2. the wave's best group (uniform; equal values: the reference's tie key decides)
    float cw_val = cg_val[0], cw_x = cg_x[0], cw_y = cg_y[0], cw_z = cg_z[0];
    int cw_k = cg_k[0];
#pragma unroll
    for (int g = 1; g < G; ++g) {
      const bool better = cg_val[g] > cw_val || (cg_val[g] == cw_val && tie_key(cg_k[g], log2s) < tie_key(cw_k, log2s));
      if (better) { cw_val = cg_val[g]; cw_k = cg_k[g]; cw_x = cg_x[g]; cw_y = cg_y[g]; cw_z = cg_z[g]; }
    }
    // 3. block arg-max over the waves' cached maxima
    if (lane == 0) {
      slot_v[r & 1][wave] = make_float4(cw_val, cw_x, cw_y, cw_z);   // one ds_write_b128 + one ds_write_b32
      slot_k[r & 1][wave] = cw_k;
    }
    __syncthreads();
    const int src = lane & (NW - 1);          // every 16-lane row holds all NW <= 16 entries
    const float4 e2 = slot_v[r & 1][src];
    const int i2 = slot_k[r & 1][src];
    const float v2 = e2.x, x2 = e2.y, y2 = e2.z, z2 = e2.w;
    const float bmax = d6_row_max16(v2);      // 4 DPP steps instead of the 6 of a full-wave maximum
    const unsigned long long tie2 = __ballot(v2 == bmax) & ((1ull << NW) - 1ull);
    int ww = __builtin_ctzll(tie2);
    if (__popcll(tie2) != 1) ww = min_key_lane(tie2, i2, log2s);
    const int old = d6_readlane_i(i2, ww);
    cx = d6_readlane_f(x2, ww);
    cy = d6_readlane_f(y2, ww);
    cz = d6_readlane_f(z2, ww);
    if (h == 0 && (r & ((1 << PSHIFT) - 1)) == part) idxs[r] = old + idx_add;   // (PSHIFT > 0: part p supplies every 2^PSHIFT-th pick)
  }
  if (D6_DBG_IS(7) && h == 0 && blockIdx.x < 64) d6_fps_clock[2 * blockIdx.x + 1] = wall_clock64();
}

#endif  // DET6D_EXPERIMENTS

}  // namespace

#ifdef DET6D_EXPERIMENTS
// timing experiments only: copies the 2 x 64 clock samples of the last DET6D_FPS_DBG=7 sampler launch to the host
extern "C" __attribute__((visibility("default"))) int det6d_dbg_fps_clock(unsigned long long *out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(d6_fps_clock), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1;
}
#endif

// fps_seq.hip: the multi-pick sampler
int det6d_fps_seq_launch(int b, int n, int m, int log2s, long long xyz_bstride, long long idx_bstride, int idx_add,
                         const float *xyz, const int *perm, int *idx, hipStream_t stream);

// Called by fps.hip's launcher for D-FPS of 16384- and 4096-point scenes with fresh min-distances.  `perm` is (B, n) int32 scratch.
// k-d order (4 x 4 cells of equal counts, lanes ordered by the tie key) of every 16384-point part src[g] of `subscenes`
// scene parts (fps_coop.hip); perm[g] = the part's points in that order
int det6d_fps_cell_sort_parts(int subscenes, int parts, int log2s, long long xyz_bstride, const float *xyz, const int *src, int *perm,
                              hipStream_t stream) {
  hipLaunchKernelGGL((cell_sort_kernel<16>), dim3(subscenes), dim3(1024), 0, stream, 16384, log2s, xyz_bstride, xyz, perm, src, parts);
  return det6d_check_launch("det6d_fps (cooperative: k-d order of the parts)");
}

int det6d_fps_cells_launch(int b, int n, int m, int log2s, long long xyz_bstride, long long idx_bstride, int idx_add,
                           const float *xyz, int *perm, int *idx, hipStream_t stream) {
  if (n != 16384 && n != 4096) return DET6D_EINVAL;
  dim3 grid(b);
  if (n == 4096) {      // 1024 lanes x 4 points
    hipLaunchKernelGGL((cell_sort_kernel<4>), grid, dim3(1024), 0, stream, n, log2s, xyz_bstride, xyz, perm);
    return det6d_fps_seq_launch(b, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm, idx, stream);
  }
  hipLaunchKernelGGL((cell_sort_kernel<16>), grid, dim3(1024), 0, stream, n, log2s, xyz_bstride, xyz, perm);
#ifdef DET6D_EXPERIMENTS
  static const int seq = det6d_env_int("DET6D_FPS_SEQ", 1);
  if (!seq) {
    static const int dbg = det6d_env_int("DET6D_FPS_DBG", 0);  // timing experiments only
    det6d_dbg_poison_lds_hook(stream);      // DET6D_DBG_POISON_LDS: fps_seq.hip
    static const unsigned hog = det6d_sampler_lds_hog(fps_skip_kernel<16, 16, 1>, 16 * 20 * 2 + 2 * 64 * 16 * 16);
    hipLaunchKernelGGL((fps_skip_kernel<16, 16, 1>), grid, dim3(1024), hog, stream, n, m, log2s, xyz_bstride, idx_bstride,
                       idx_add, dbg, xyz, perm, idx);
    return det6d_check_launch("det6d_fps (wave skip)");
  }
#endif
  return det6d_fps_seq_launch(b, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm, idx, stream);
}

// the score-weighted form: k-d pre-pass (a permutation of the scene: the weights play no part in it) + fps_seq_w_kernel
int det6d_fps_seq_w_launch(int b, int n, int m, int log2s, long long xyz_bstride, long long idx_bstride, int idx_add,
                           const float *xyz, const int *perm, int *idx, const float *weights, long long w_bstride, float gamma,
                           int w_is_score, int *flags, hipStream_t stream);      // fps_seq.hip
int det6d_fps_cells_w_launch(int b, int n, int m, int log2s, long long xyz_bstride, long long idx_bstride, int idx_add,
                             const float *xyz, int *perm, int *idx, const float *weights, long long w_bstride, float gamma,
                             int w_is_score, hipStream_t stream) {
  if (n != 16384 && n != 4096) return DET6D_EINVAL;
  if (n == 4096) hipLaunchKernelGGL((cell_sort_kernel<4>), dim3(b), dim3(1024), 0, stream, n, log2s, xyz_bstride, xyz, perm);
  else hipLaunchKernelGGL((cell_sort_kernel<16>), dim3(b), dim3(1024), 0, stream, n, log2s, xyz_bstride, xyz, perm);
  return det6d_fps_seq_w_launch(b, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm, idx, weights, w_bstride, gamma, w_is_score,
                                perm, stream);
}

