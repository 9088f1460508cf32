// fps_seq.hip — multi-pick farthest point sampling (D-FPS) for gfx950: bit for bit the picks of
// farthest_point_sampling_kernel (core/pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:101-222, tie order of its
// shared-memory tree :94-99,159-216), several picks per barrier round.
//
// The wave-skip sampler (fps_cells.hip) pays  rescan -> LDS slot -> s_barrier -> block arg-max  for EVERY pick: 0.86 us x 4095.
// Here a round still has that shape, but the waves publish their top kCand points instead of their maximum, and one wave
// (the sequencer, wave 0) decides AS MANY picks as those lists allow before the next round of rescans:
//
//   Record of a wave (region = the 1024 Morton-consecutive points it holds in registers), rewritten whenever a pick may
//   have changed the region: its top (up to) kCand points c_1 > c_2 > .. in the reference's order (value descending, then
//   tie key ascending) as {value, index, x, y, z}.  Every other point of the region is ordered AFTER the last candidate,
//   and min-distances only decrease.
//   Sequencer (lane 4w + s = candidate s of wave w): keeps the CURRENT value cv of every candidate exact by applying each
//   pick it makes with the scan's own distance expression.  For a region,  X = its best current candidate;  its maximum is
//   exactly X  iff  X is ordered before-or-at the record's last candidate as it was; otherwise the maximum is unknown but
//   <= that candidate's old value.  The next pick is the best exact X provided every unknown region's bound is strictly
//   below it; otherwise the round ends.  At the start of a round every record is fresh, so the first pick always goes
//   through: a round makes >= 1 pick, typically 4-6 (tests/models/fps_lookahead.py, `greedy` schedule, is this rule as an
//   executable model; tests/test_fps_lookahead_model.py checks it against plain FPS on ties, duplicates, lattices).
//   Owners (all 16 waves): test the round's picks against their bounding box (a pick at least sqrt(current maximum) away
//   changes nothing: the floating-point box distance is a lower bound of every distance the scan would compute,
//   fps_cells.hip), apply the ones that may matter in ONE pass and rewrite their record.
// Two barriers per round; between them only wave 0 works (the other 15 sleep at the barrier: no issue slots taken from
// co-resident GEMM waves).
#include "common.h"

namespace {

typedef unsigned long long u64;
typedef float sq_f32x2 __attribute__((ext_vector_type(2)));

constexpr int kCandMax = 4;              // candidates per record = sequencer lanes per region (2 or 4)
constexpr int kWaves = 16, kSlots = 16;  // 16 x 64 x 16 = 16384 points
constexpr int kMaxPicks = 16;            // picks per round at most
static_assert(kWaves * kCandMax == 64, "one sequencer lane per candidate");

__device__ __forceinline__ unsigned sq_bitrev_bits(unsigned v, int bits) {
  return bits == 0 ? 0u : (__builtin_bitreverse32(v) >> (32 - bits));
}
// order key of point k under the reference's tie rule (smaller wins): (bitrev_{log2 S}(k mod S), k)
__device__ __forceinline__ unsigned sq_tie_key(int k, int log2s) {
  return (sq_bitrev_bits((unsigned)k & ((1u << log2s) - 1u), log2s) << (32 - log2s)) | ((unsigned)k >> log2s);
}
// lane holding the smallest key among the lanes of `cand` (tie path only)
__device__ __forceinline__ int sq_min_key_lane(u64 cand, unsigned key) {
  const int lane = threadIdx.x & 63;
  const bool mine = (cand >> lane) & 1ull;
  const unsigned k = mine ? key : 0xFFFFFFFFu;
  unsigned m = k;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)m, off);
    m = o < m ? o : m;
  }
  return __builtin_ctzll(__ballot(mine && k == m));
}

// slot ws (wave-uniform) of this lane's coordinate registers: scalar binary search down to the statically indexed slot
template <int LO, int HI, int N>
__device__ __forceinline__ void sq_select(int ws, const float (&px)[N], const float (&py)[N], const float (&pz)[N],
                                          float &x, float &y, float &z) {
  if constexpr (HI - LO == 1) {
    x = px[LO]; y = py[LO]; z = pz[LO];
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (ws < MID) sq_select<LO, MID>(ws, px, py, pz, x, y, z);
    else sq_select<MID, HI>(ws, px, py, pz, x, y, z);
  }
}

// max over the four lanes of every quad (lanes 4q .. 4q+3), in all four lanes; two values at once (the DPP steps interleave)
__device__ __forceinline__ void sq_quad_max2(float a, float b, float &ra, float &rb) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      : "=&v"(ra), "=&v"(rb)
      : "v"(a), "v"(b));
}

// max over the four lanes of every quad of one value
__device__ __forceinline__ float sq_quad_max(float a) {
  float r;
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      : "=&v"(r)
      : "v"(a));
  return r;
}

// max over the lanes of every PAIR (lanes 2p, 2p+1)
__device__ __forceinline__ float sq_pair_max(float a) {
  float r;
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      : "=&v"(r)
      : "v"(a));
  return r;
}
template <int K>
__device__ __forceinline__ float sq_group_max(float a) {
  if constexpr (K == 2) return sq_pair_max(a);
  else return sq_quad_max(a);
}

// max over the 64 lanes of an unsigned value (uniform result)
__device__ __forceinline__ unsigned sq_wave_max_u32(unsigned v) {
  unsigned t;
  asm volatile(
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      : "=&v"(t)
      : "v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)t, 63);
}

// max over the 64 lanes of two values at once (uniform results)
__device__ __forceinline__ void sq_wave_max2(float a, float b, float &ra, float &rb) {
  float ta, tb;
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      : "=&v"(ta), "=&v"(tb)
      : "v"(a), "v"(b));
  ra = d6_readlane_f(ta, 63);
  rb = d6_readlane_f(tb, 63);
}

// the published record of a wave: candidate i of wave w at index 4w + i (= the sequencer lane that reads it)
struct SqRecords {
  float v[64], x[64], y[64], z[64];
  int k[64];
  int nc[kWaves];
};

// min-distances of this wave's points against one more pick; no arg-max bookkeeping (the sq_rescan that closes the batch
// does it once for all the picks)
template <int SG>
__device__ __forceinline__ void sq_apply(float cx, float cy, float cz, const float (&px)[SG], const float (&py)[SG],
                                         const float (&pz)[SG], float (&pt)[SG]) {
  const sq_f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
  for (int q = 0; q < SG / 2; ++q) {
    const int s0 = 2 * q;
    const sq_f32x2 dx = sq_f32x2{px[s0], px[s0 + 1]} - c2x;
    const sq_f32x2 dy = sq_f32x2{py[s0], py[s0 + 1]} - c2y;
    const sq_f32x2 dz = sq_f32x2{pz[s0], pz[s0 + 1]} - c2z;
    sq_f32x2 d = dy * dy;
    d = __builtin_elementwise_fma(dx, dx, d);
    d = __builtin_elementwise_fma(dz, dz, d);
    pt[s0] = d6_vmin(d[0], pt[s0]);
    pt[s0 + 1] = d6_vmin(d[1], pt[s0 + 1]);
  }
}

// Apply the pick (cx, cy, cz) to this wave's points, extract the record and publish it.  Per lane the best value (+ slot) and
// the second best are tracked in the scan (med3); the candidates are taken one by one as the best lane head under the order,
// and the lane that holds a candidate writes it to the record itself.  A lane knows only its two best points, so the list
// ends with the first candidate that is a lane's SECOND (what is left in that lane is ordered after it, but not necessarily
// after later heads).  Returns the region's maximum.
template <int SG, int kCand>
__device__ __forceinline__ float sq_rescan(float cx, float cy, float cz, int log2s, const float (&px)[SG], const float (&py)[SG],
                                           const float (&pz)[SG], float (&pt)[SG], const unsigned short *korig_w, SqRecords &rec,
                                           int wave) {
  const int lane = threadIdx.x & 63;
  float best = -1.0f, sec = -1.0f;
  int bs = 0;
  const sq_f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
  auto visit = [&](int j, float d) {
    const float t = d6_vmin(d, pt[j]);
    pt[j] = t;
    sec = __builtin_amdgcn_fmed3f(best, sec, t);          // second best so far (uses the OLD best)
    const bool up = t > best;
    bs = up ? j : bs;
    best = d6_vmax(best, t);
  };
#pragma unroll
  for (int q = 0; q < SG / 2; ++q) {
    const int s0 = 2 * q;
    const sq_f32x2 dx = sq_f32x2{px[s0], px[s0 + 1]} - c2x;
    const sq_f32x2 dy = sq_f32x2{py[s0], py[s0 + 1]} - c2y;
    const sq_f32x2 dz = sq_f32x2{pz[s0], pz[s0 + 1]} - c2z;
    sq_f32x2 d = dy * dy;
    d = __builtin_elementwise_fma(dx, dx, d);
    d = __builtin_elementwise_fma(dz, dz, d);
    visit(s0, d[0]);
    visit(s0 + 1, d[1]);
  }
  static_assert(SG % 2 == 0, "slot pairs");
  auto second_slot = [&]() -> int {                      // lowest slot != bs holding the lane's second value
    // (rare path.  The value searched for goes through an opaque asm so that the 16-step search stays inside the branch that
    // needs it: left alone, the compiler hoists it in front of the candidate loop of EVERY rescan)
    float target = sec;
    int skip = bs;
    asm volatile("" : "+v"(target), "+v"(skip));
    int ss = 0;
#pragma unroll
    for (int j = SG - 1; j >= 0; --j) ss = (pt[j] == target && j != skip) ? j : ss;
    return ss;
  };

  int taken = 0;            // this lane's best has been taken
  float head = best;
  float cmax = 0.f;
  int nc = 0;
#pragma nounroll
  for (int i = 0; i < kCand; ++i) {
    const float wm = d6_wave_max(head);
    if (i == 0) cmax = wm;
    const u64 tie = __ballot(head == wm);
    int wl = __builtin_ctzll(tie);
    int ss = 0;
    bool have_ss = false;
    if (__popcll(tie) != 1) {                              // equal heads: the reference's key decides
      if (__ballot(taken != 0 && head == wm) != 0ull) { ss = second_slot(); have_ss = true; }
      const int hs = taken ? ss : bs;
      wl = sq_min_key_lane(tie, sq_tie_key((int)korig_w[lane * SG + hs], log2s));
    }
    const int wtk = d6_readlane_i(taken, wl);
    int ws;
    if (wtk == 0) {
      ws = d6_readlane_i(bs, wl);
    } else {
      if (!have_ss) ss = second_slot();
      ws = d6_readlane_i(ss, wl);
    }
    if (lane == wl) {                                      // the holder writes its candidate
      float x, y, z;
      sq_select<0, SG>(ws, px, py, pz, x, y, z);
      const int o = wave * kCandMax + i;
      rec.v[o] = wm;
      rec.k[o] = (int)korig_w[lane * SG + ws];
      rec.x[o] = x; rec.y[o] = y; rec.z[o] = z;
      taken = 1;
      head = sec;
    }
    nc = i + 1;
    if (wtk != 0) break;                                   // a lane is exhausted: the list ends here
  }
  if (lane == 0) rec.nc[wave] = nc;
  return cmax;
}

#ifdef DET6D_EXPERIMENTS
// scripts/experiments only: counters of workgroup 0 of the last launch (det6d_dbg_fps_seq_stats): 0 rounds, 1 picks,
// 2 rescans (all waves), 3 extra applies (all waves), 4 rounds that ended on an unknown region, 5 total cycles,
// 6 / 7 cycles of waves 0 / 5 in phase A, 8 cycles wave 0 waits at the first barrier, 9 cycles of phase B
__device__ unsigned long long d6_fps_seq_stats[16];
#define SQ_STAT(i, v) do { const unsigned long long sv_ = (unsigned long long)(v); if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicAdd(&sq_stats_lds[i], sv_); } while (0)
#else
#define SQ_STAT(i, v) do { } while (0)
#endif

// One workgroup of 16 waves per scene.  `perm`: the scene's Morton permutation whose lane groups of 16 consecutive positions
// are ordered by tie key (fps_cells.hip: cell_sort_kernel + skip_group_order_kernel<16>).
template <int kCand>
__global__ __launch_bounds__(1024) void fps_seq_kernel(int n, int m, int log2s, long long xyz_bstride, long long idx_bstride,
                                                       int idx_add, const float *__restrict__ xyz,
                                                       const int *__restrict__ perm, int *__restrict__ idxs, int max_picks) {
  constexpr int SG = kSlots;
  __shared__ unsigned short korig[64 * kWaves * kSlots];   // sorted position -> original index
  __shared__ SqRecords rec;
  __shared__ float pick_x[kMaxPicks], pick_y[kMaxPicks], pick_z[kMaxPicks];
  __shared__ int pick_n;
#ifdef DET6D_EXPERIMENTS
  __shared__ unsigned long long sq_stats_lds[16];
  if (threadIdx.x < 16) sq_stats_lds[threadIdx.x] = 0ull;
  const long long t_begin = clock64();
#endif
  const int h = threadIdx.x, lane = h & 63;
  const int wave = __builtin_amdgcn_readfirstlane(h >> 6);
  xyz += (size_t)blockIdx.x * xyz_bstride;
  perm += (size_t)blockIdx.x * n;
  idxs += (size_t)blockIdx.x * idx_bstride;
  unsigned short *korig_w = korig + (size_t)wave * 64 * SG;

  float px[SG], py[SG], pz[SG], pt[SG];
  float lox, loy, loz, hix, hiy, hiz;
  {
    float ax = 3.0e38f, ay = 3.0e38f, az = 3.0e38f, bx = -3.0e38f, by = -3.0e38f, bz = -3.0e38f;
#pragma unroll
    for (int j = 0; j < SG; ++j) {
      const int pos = (wave * 64 + lane) * SG + j;
      const int k = perm[pos];
      korig[pos] = (unsigned short)k;
      px[j] = xyz[(size_t)k * 3 + 0];
      py[j] = xyz[(size_t)k * 3 + 1];
      pz[j] = xyz[(size_t)k * 3 + 2];
      asm volatile("" : "+v"(px[j]), "+v"(py[j]), "+v"(pz[j]));
      pt[j] = 1e10f;
      ax = d6_vmin(ax, px[j]); bx = d6_vmax(bx, px[j]);
      ay = d6_vmin(ay, py[j]); by = d6_vmax(by, py[j]);
      az = d6_vmin(az, pz[j]); bz = d6_vmax(bz, pz[j]);
    }
    lox = d6_wave_min(ax); hix = d6_wave_max(bx);
    loy = d6_wave_min(ay); hiy = d6_wave_max(by);
    loz = d6_wave_min(az); hiz = d6_wave_max(bz);
  }
  float cmax = __builtin_inff();
  // pick 0 is point 0 (sampling_gpu.cu:131-133): the first round's only pick; every wave's box test passes against +inf
  if (h == 0) { pick_x[0] = xyz[0]; pick_y[0] = xyz[1]; pick_z[0] = xyz[2]; pick_n = 1; idxs[0] = idx_add; }
  __syncthreads();

  int r = 1;                                              // picks made so far
  while (true) {
    // ---- A. every wave: the picks of the last round against its points
#ifdef DET6D_EXPERIMENTS
    const long long ta0 = clock64();
#endif
    {
      const int np = pick_n;
      const float sx = pick_x[lane & (kMaxPicks - 1)], sy = pick_y[lane & (kMaxPicks - 1)], sz = pick_z[lane & (kMaxPicks - 1)];
      const float gx = fmaxf(0.f, fmaxf(lox - sx, sx - hix));
      const float gy = fmaxf(0.f, fmaxf(loy - sy, sy - hiy));
      const float gz = fmaxf(0.f, fmaxf(loz - sz, sz - hiz));
      const float lb = d6_sqdist(gx, gy, gz);
      // (tested against the maximum BEFORE the batch: min-distances only decrease, the test stays conservative)
      u64 need = __ballot(lane < np && !(lb >= cmax));
      if (need != 0ull) {
        while (need & (need - 1ull)) {
          const int i = __builtin_ctzll(need);
          need &= need - 1ull;
          sq_apply<SG>(d6_readlane_f(sx, i), d6_readlane_f(sy, i), d6_readlane_f(sz, i), px, py, pz, pt);
          SQ_STAT(3, 1);
        }
        const int i = __builtin_ctzll(need);
        SQ_STAT(2, 1);
        cmax = sq_rescan<SG, kCand>(d6_readlane_f(sx, i), d6_readlane_f(sy, i), d6_readlane_f(sz, i), log2s, px, py, pz, pt, korig_w, rec, wave);
      }
    }
    if (r >= m) break;                                    // (uniform over the workgroup: r is advanced by pick_n everywhere)
#ifdef DET6D_EXPERIMENTS
    const long long ta1 = clock64();
    if (wave == 0) SQ_STAT(6, ta1 - ta0);
    if (wave == 5) SQ_STAT(7, ta1 - ta0);
#endif
    __syncthreads();                                      // records complete
#ifdef DET6D_EXPERIMENTS
    const long long tb0 = clock64();
    if (wave == 0) SQ_STAT(8, tb0 - ta1);
#endif
    // ---- B. wave 0: as many picks as the records allow
    if (wave == 0) {
      SQ_STAT(0, 1);
      constexpr int LOG2K = kCand == 2 ? 1 : 2;
      const bool live = lane < kWaves * kCand;
      const int region = live ? lane >> LOG2K : 0;
      const int nc = rec.nc[region];
      const int slot = lane & (kCand - 1);
      const int o = region * kCandMax + slot;             // records keep the stride of four
      float cv = live && slot < nc ? rec.v[o] : -1.0f;    // current value of this candidate (-1: no candidate in this slot)
      const float qx = rec.x[o], qy = rec.y[o], qz = rec.z[o];
      const int kidx = rec.k[o];
      const bool is_last = live && slot == nc - 1;
      const float bound_v = sq_group_max<kCand>(is_last ? cv : -2.0f);   // value of the record's last candidate (group-uniform)
      int j = 0;
      const int jmax = min(max_picks, m - r);
      for (; j < jmax; ++j) {
        // The region's maximum is exactly its best candidate X iff X is ordered before-or-at the record's last candidate as
        // it was: some candidate's value above the bound, or the last candidate itself untouched (then X is that candidate
        // or one ordered before it; an equal value reached by coincidence counts as unknown: the round ends, the region is
        // rescanned — the pick that lowered its last candidate passes its box test — and is fresh in the next one).
        const float pe = (cv > bound_v || (is_last && cv == bound_v)) ? 1.0f : 0.0f;
        const bool exact = sq_group_max<kCand>(pe) != 0.0f;
        // ONE reduction decides both questions: key = 2 * bits(value) + (1 for the bound of an unknown region, 0 for a
        // candidate of an exact one); values are >= +0, so their bits order like the values and fit 31 bits; an empty slot
        // (-1) has key 0, below the key 2 of a zero value.  The largest key wins: odd = an unknown region may hold the maximum (ties go to it): the round ends.
        const float val = exact ? cv : bound_v;
        const unsigned key = val < 0.f ? 0u : ((__builtin_bit_cast(unsigned, val) << 1) | (exact ? 0u : 1u)) + 2u;
        const unsigned best = sq_wave_max_u32(key);
        if (best & 1u) { SQ_STAT(4, 1); break; }
        const u64 tie = __ballot(key == best);
        int wl = __builtin_ctzll(tie);
        if (__popcll(tie) != 1) wl = sq_min_key_lane(tie, sq_tie_key(kidx, log2s));
        if (lane == wl) {                                  // the holder publishes the pick
          pick_x[j] = qx; pick_y[j] = qy; pick_z[j] = qz;
          idxs[r + j] = kidx + idx_add;
        }
        const float sx = d6_readlane_f(qx, wl), sy = d6_readlane_f(qy, wl), sz = d6_readlane_f(qz, wl);
        cv = d6_vmin(cv, d6_sqdist(qx - sx, qy - sy, qz - sz));   // (an empty slot stays at -1)
      }
      if (lane == 0) pick_n = j;
      SQ_STAT(1, j);
#ifdef DET6D_EXPERIMENTS
      SQ_STAT(9, clock64() - tb0);
#endif
    }
    __syncthreads();                                      // picks published
    r += pick_n;
  }
#ifdef DET6D_EXPERIMENTS
  if (wave == 0) {
    SQ_STAT(5, clock64() - t_begin);
    __builtin_amdgcn_s_waitcnt(0);
    if (blockIdx.x == 0 && lane < 16) d6_fps_seq_stats[lane] = sq_stats_lds[lane];
  }
#endif
}

}  // namespace

#ifdef DET6D_EXPERIMENTS
extern "C" __attribute__((visibility("default"))) int det6d_dbg_fps_seq_stats(unsigned long long *out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(d6_fps_seq_stats), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

// Called by fps_cells.hip's launcher after the Morton sort and the lane-group ordering (groups of 16 positions).
int det6d_fps_seq_launch(int b, int n, int m, int log2s, int regions_per_wave, long long xyz_bstride, long long idx_bstride,
                         int idx_add, const float *xyz, const int *perm, int *idx, hipStream_t stream) {
  if (n != 16384) return DET6D_EINVAL;
  (void)regions_per_wave;
  static const int max_picks_env = det6d_env_int("DET6D_FPS_SEQ_PICKS", kMaxPicks);
  static const int cands = det6d_env_int("DET6D_FPS_SEQ_CANDS", 4);
  const int max_picks = max_picks_env < 1 ? 1 : max_picks_env > kMaxPicks ? kMaxPicks : max_picks_env;
  if (cands == 2)
    hipLaunchKernelGGL(fps_seq_kernel<2>, dim3(b), dim3(1024), 0, stream, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm, idx, max_picks);
  else
    hipLaunchKernelGGL(fps_seq_kernel<4>, dim3(b), dim3(1024), 0, stream, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm, idx, max_picks);
  return det6d_check_launch("det6d_fps (multi-pick)");
}
