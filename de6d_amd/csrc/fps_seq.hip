// fps_seq.hip — multi-pick farthest point sampling (D-FPS) for gfx950: bit for bit the picks of
// farthest_point_sampling_kernel (core/pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:101-222, tie order of its
// shared-memory tree :94-99,159-216), several picks per barrier round.
//
// The 16384- and 4096-point D-FPS of the library (pre-pass and launcher: fps_cells.hip).  The one-pick wave-skip sampler of rounds 2-3
// (fps_cells.hip, experiments build) pays  rescan -> LDS slot -> s_barrier -> block arg-max  for EVERY pick: 0.84 us x 4095.
// Here a round still has that shape, but the waves publish their top kCand points instead of their maximum, and one wave
// (the sequencer, wave 0) decides AS MANY picks as those lists allow before the next round of rescans:
//
//   Record of a wave (region = the k-d cell of 1024 / 256 points it holds in registers), rewritten whenever a pick may
//   have changed the region: its top (up to) kCand points c_1 > c_2 > .. in the reference's order (value descending, then
//   tie key ascending) as {value, index, x, y, z}.  Every other point of the region is ordered AFTER the last candidate,
//   and min-distances only decrease.
//   Sequencer (lane 4w + s = candidate s of wave w): keeps the CURRENT value cv of every candidate exact by applying each
//   pick it makes with the scan's own distance expression.  For a region,  X = its best current candidate;  its maximum is
//   exactly X  iff  X is ordered before-or-at the record's last candidate as it was; otherwise the maximum is unknown but
//   <= that candidate's old value.  The next pick is the best exact X provided every unknown region's bound is strictly
//   below it; otherwise the round ends.  At the start of a round every record is fresh, so the first pick always goes
//   through: a round makes >= 1 pick, 8.5 on average (tests/models/fps_lookahead.py, `greedy` schedule, is this rule as an
//   executable model; tests/test_fps_lookahead_model.py checks it against plain FPS on ties, duplicates, lattices).
//   Owners (all 16 waves): test the round's picks against their bounding box (a pick at least sqrt(current maximum) away
//   changes nothing: the floating-point box distance is a lower bound of every distance the scan would compute,
//   fps_cells.hip), apply the ones that may matter in ONE pass and rewrite their record.
// Two barriers per round; between them only wave 0 works (the other 15 sleep at the barrier: no issue slots taken from
// co-resident GEMM waves).
#include <stdio.h>
#include "fps_multi.h"

namespace {

#ifdef DET6D_EXPERIMENTS
// scripts/experiments only: counters of workgroup 0 of the last launch (det6d_dbg_fps_seq_stats): 0 rounds, 1 picks,
// 2 rescans (all waves), 3 extra applies (all waves), 4 rounds that ended on an unknown region, 5 total cycles,
// 6 / 7 cycles of waves 0 / 5 in phase A, 8 cycles wave 0 waits at the first barrier, 9 cycles of phase B
__device__ unsigned long long d6_fps_seq_stats[16];
#define SQ_STAT(i, v) do { const unsigned long long sv_ = (unsigned long long)(v); if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicAdd(&sq_stats_lds[i], sv_); } while (0)
#else
#define SQ_STAT(i, v) do { } while (0)
#endif

// One workgroup of 16 waves per scene.  `perm`: the scene's k-d permutation whose lane groups of SG consecutive positions
// are ordered by tie key (fps_cells.hip: cell_sort_kernel).
template <int kCand, int SG = kSlots>
__global__ __launch_bounds__(1024) void fps_seq_kernel(int n, int m, int log2s, long long xyz_bstride, long long idx_bstride,
                                                       int idx_add, const float *__restrict__ xyz,
                                                       const int *__restrict__ perm, int *__restrict__ idxs, int max_picks, int depth_add) {
  // SG points per lane: 16 for 16384-point scenes, 4 for 4096-point ones (n = 1024 SG)
  __shared__ unsigned short korig[64 * kWaves * SG];       // sorted position -> original index
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
  sq_hide_lane_duplicates<SG>(px, py, pz, pt);
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
      // (the first round always rescans: a wave's record is written by its first rescan, and a point 0 with a huge or
      // infinite coordinate would make lb overflow to +inf == cmax)
      u64 need = __ballot(lane < np && (r == 1 || !(lb >= cmax)));
      if (need != 0ull) {
        while (need & (need - 1ull)) {
          const int i = __builtin_ctzll(need);
          need &= need - 1ull;
          sq_apply<SG>(d6_readlane_f(sx, i), d6_readlane_f(sy, i), d6_readlane_f(sz, i), px, py, pz, pt);
          SQ_STAT(3, 1);
        }
        const int i = __builtin_ctzll(need);
        SQ_STAT(2, 1);
        // list depth: short lists after a round that made a single pick (the first rounds — every pick reaches every region —
        // make one each: top-4 lists would be extracted for nothing), full lists otherwise.  (Tying the depth to the number
        // of picks more closely — picks + 1 — is a trap: short lists make short rounds make short lists; ray-cast scenes stayed
        // there for long stretches, 2.03 -> 2.69 ms.)
        cmax = sq_rescan<SG, kCand>(d6_readlane_f(sx, i), d6_readlane_f(sy, i), d6_readlane_f(sz, i), log2s, px, py, pz, pt, korig_w, rec, wave,
                                    np <= depth_add ? min(kCand, 2) : kCand);
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
      // Keys: an exact candidate of value v has the EVEN key 2 bits(v) + 2 (values are >= +0: their bits order like the
      // values and fit 31 bits), a region whose maximum is unknown the ODD key ub = 2 bits(bound) + 3, so that an unknown
      // region outranks an exact candidate of the same value (it may hold a point of that value with a better tie key);
      // 0 = nothing.  A region's maximum is exactly its best candidate X iff X is ordered before-or-at the record's last
      // candidate L as it was — (value, tie key) of X now against (value, tie key) of L then: a value above the bound, or
      // the bound itself with a tie key not after L's (L untouched is the plain case; a candidate that reaches the bound
      // by coincidence with a later key is NOT such an X).  Every lane can tell that of its own candidate:
      //   thr = the smallest key with which THIS candidate is such an X: ub - 1 (the bound itself) if its tie key is not
      //         after L's, else ub (above the bound: even against odd, >= is >); all ones for an empty slot;
      //   alt = what the lane puts forward otherwise: the region's unknown key from L's lane, nothing from the others.
      // One wave-wide max then answers both questions: an X above the bound beats its own region's unknown key, and an
      // untouched L withholds it.  (A touched L still puts the unknown key forward when another candidate sits exactly on
      // the bound: conservative, the round ends early.  At the start of a round every L is untouched: no unknown key, the
      // first decision always goes through.)
      const unsigned ub = ((__builtin_bit_cast(unsigned, bound_v) << 1) | 1u) + 2u;
      const bool alive = live && slot < nc;
      const unsigned ntk = ~sq_tie_key(kidx, log2s);                                  // larger = earlier in the order
      const unsigned ntk_last = sq_group_max_u32<kCand>(is_last ? ntk : 0u);
      const unsigned thr = !alive ? 0xFFFFFFFFu : ntk >= ntk_last ? ub - 1u : ub;
      const unsigned alt = is_last ? ub : 0u;
      int j = 0;
      const int jmax = min(max_picks, m - r);
      // (round 6) the picks are PUBLISHED after the loop: lane j remembers WHICH lane's candidate pick j was (one select per
      // decision) and fetches its coordinates / index with four cross-lane reads once per round, instead of the holder's
      // exec-masked LDS / global stores with their 64-bit address inside the chain of every decision: 0.440 -> 0.409 us per
      // pick on the benchmark scenes with the first form of this (values kept per lane)
      int mwl = 0;       // lane j: the lane whose candidate pick j was (its coordinates and index are constants of the round)
      // ONE loop exit: the decision that ends the round (an odd best key) still runs the rest of the body — its bogus winner
      // changes cv and lane j's note, which nobody reads again: cv is rebuilt from the records next round and lane j >= the
      // final j publishes nothing — instead of leaving from the middle: with two exits the backend's structurizer turns both
      // uniform branches into mask arithmetic (s_cselect / s_and / s_xor / s_andn2: nine scalar instructions per decision)
      {
        bool go;
        do {
          const unsigned ekey = (__builtin_bit_cast(unsigned, cv) << 1) + 2u;
          const unsigned key = ekey >= thr ? ekey : alt;
          // the largest key wins: odd = an unknown region may hold the maximum (ties go to it): the round ends
          const unsigned best = sq_wave_max_u32(key);
          const u64 tie = __ballot(key == best);
          int wl = __builtin_ctzll(tie);
          if (__popcll(tie) != 1) wl = sq_min_key_lane(tie, ~ntk);
          const float sx = d6_readlane_f(qx, wl), sy = d6_readlane_f(qy, wl), sz = d6_readlane_f(qz, wl);
          mwl = lane == j ? wl : mwl;
          cv = d6_vmin(cv, d6_sqdist(qx - sx, qy - sy, qz - sz));   // (an empty slot stays at -1)
          go = (best & 1u) == 0u;
          j += go ? 1 : 0;
#ifdef DET6D_EXPERIMENTS
          if (!go) SQ_STAT(4, 1);
#endif
        } while (go && j < jmax);
      }
      {
        const float mx = __shfl(qx, mwl), my = __shfl(qy, mwl), mz = __shfl(qz, mwl);
        const int mk = __shfl(kidx, mwl);
        if (lane < j) {
          pick_x[lane] = mx; pick_y[lane] = my; pick_z[lane] = mz;
          idxs[r + lane] = mk + idx_add;
        }
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

// ---- round 6: the SCORE-WEIGHTED sampler (S-FPS) in the multi-pick form ------------------------------------------------------
// furthest_point_sampling_weights_kernel (sampling_gpu.cu:419-540): pick 0 = arg-max of the weights, pick j > 0 = arg-max of
// float(double(min-distance) * max(double(w), 1e-12)), the reference's tie order.  While every weight of the scene is >= 1e-12
// (and none is NaN) that score is the IEEE fp32 product t * w (fps.hip, "S-FPS in exact fp32"); a scene that holds a smaller
// weight flags itself in the first word of its scratch and leaves: the guarded memory-resident launch behind this one samples
// it with the double expression (same protocol as fps_fat_kernel<.., FASTW>).
// Same rounds as fps_seq_kernel: records hold a region's top candidates BY SCORE (+ their min-distance and weight), the
// sequencer keeps every candidate's min-distance exact and rebuilds its score per decision (one more multiply in the chain),
// a region whose best candidate has sunk below the record's last one is unknown with that candidate's old SCORE as the bound
// (scores only decrease: the weights are constants), and the owners' box test compares a pick's distance to the box with the
// region's maximal MIN-DISTANCE (a pick that far away lowers no min-distance, hence no score).
template <int kCand, int SG>
__global__ __launch_bounds__(1024) void fps_seq_w_kernel(int n, int m, int log2s, long long xyz_bstride, long long idx_bstride,
                                                         int idx_add, const float *__restrict__ xyz, const int *__restrict__ perm,
                                                         int *__restrict__ idxs, int max_picks, int depth_add,
                                                         const float *__restrict__ weights, long long w_bstride, float gamma,
                                                         int w_is_score, int *__restrict__ flags) {
  __shared__ unsigned short korig[64 * kWaves * SG];       // sorted position -> original index
  __shared__ SqRecords rec;
  __shared__ SqRecordsW recw;
  __shared__ float pick_x[kMaxPicks], pick_y[kMaxPicks], pick_z[kMaxPicks];
  __shared__ int pick_n;
  __shared__ float w0_val[kWaves], w0_x[kWaves], w0_y[kWaves], w0_z[kWaves];
  __shared__ unsigned w0_key[kWaves];
  __shared__ int small_any;
  const int h = threadIdx.x, lane = h & 63;
  const int wave = __builtin_amdgcn_readfirstlane(h >> 6);
  xyz += (size_t)blockIdx.x * xyz_bstride;
  perm += (size_t)blockIdx.x * n;
  idxs += (size_t)blockIdx.x * idx_bstride;
  weights += (size_t)blockIdx.x * w_bstride;
  flags += (size_t)blockIdx.x * n;                         // the scene's scratch (= its permutation: read below, then free)
  unsigned short *korig_w = korig + (size_t)wave * 64 * SG;
  if (h == 0) small_any = 0;

  float px[SG], py[SG], pz[SG], pt[SG], pw[SG];
  float lox, loy, loz, hix, hiy, hiz;
  bool small = false;
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
      float w = weights[k];
      if (w_is_score) w = d6_sigmoid_powf(w, gamma);
      pw[j] = w;
      small = small || !((double)w >= 1e-12);
      asm volatile("" : "+v"(px[j]), "+v"(py[j]), "+v"(pz[j]), "+v"(pw[j]));
      pt[j] = 1e10f;
      ax = d6_vmin(ax, px[j]); bx = d6_vmax(bx, px[j]);
      ay = d6_vmin(ay, py[j]); by = d6_vmax(by, py[j]);
      az = d6_vmin(az, pz[j]); bz = d6_vmax(bz, pz[j]);
    }
    lox = d6_wave_min(ax); hix = d6_wave_max(bx);
    loy = d6_wave_min(ay); hiy = d6_wave_max(by);
    loz = d6_wave_min(az); hiz = d6_wave_max(bz);
  }
  __syncthreads();                                         // small_any initialised; every thread has read its part of the permutation
  if (__ballot(small) != 0ull && lane == 0) small_any = 1;
  // ---- pick 0: arg-max of the weights (sampling_gpu.cu:452-455), ties by the reference's key.  A lane's slots are in key
  // order, so its first maximum is its smallest key.
  {
    float bw = pw[0];
    int bs0 = 0;
#pragma unroll
    for (int j = 1; j < SG; ++j) {
      const bool up = pw[j] > bw;
      bs0 = up ? j : bs0;
      bw = up ? pw[j] : bw;
    }
    const float wm = d6_wave_max(small ? 0.f : bw);        // (a flagged scene leaves below: whatever this computes is dropped)
    const u64 tie = __ballot(bw == wm);
    int wl = tie ? __builtin_ctzll(tie) : 0;
    const unsigned mykey = sq_tie_key((int)korig_w[lane * SG + bs0], log2s);
    if (__popcll(tie) > 1) wl = sq_min_key_lane(tie, mykey);
    if (lane == wl) {
      float x, y, z;
      sq_select<0, SG>(bs0, px, py, pz, x, y, z);
      w0_val[wave] = bw; w0_key[wave] = mykey; w0_x[wave] = x; w0_y[wave] = y; w0_z[wave] = z;
    }
  }
  __syncthreads();
  if (h == 0) flags[0] = small_any;                        // (the permutation has been consumed: first barrier above)
  if (small_any) return;                                   // sampled by the guarded exact-double launch behind this one
  if (wave == 0) {
    const bool mine = lane < kWaves;
    const float v = mine ? w0_val[lane & (kWaves - 1)] : -1.0f;
    const unsigned key = mine ? w0_key[lane & (kWaves - 1)] : 0xFFFFFFFFu;
    const float vm = d6_wave_max(v);
    const u64 tie = __ballot(mine && v == vm);
    int wl = __builtin_ctzll(tie);
    if (__popcll(tie) > 1) wl = sq_min_key_lane(tie, key);
    if (lane == wl) {
      pick_x[0] = w0_x[lane]; pick_y[0] = w0_y[lane]; pick_z[0] = w0_z[lane];
      pick_n = 1;
      idxs[0] = sq_tie_key_point(key, log2s) + idx_add;
    }
  }
  sq_hide_lane_duplicates_w<SG>(px, py, pz, pw, pt);
  float cmax = __builtin_inff();
  __syncthreads();

  int r = 1;                                              // picks made so far
  while (true) {
    // ---- A. every wave: the picks of the last round against its points
    {
      const int np = pick_n;
      const float sx = pick_x[lane & (kMaxPicks - 1)], sy = pick_y[lane & (kMaxPicks - 1)], sz = pick_z[lane & (kMaxPicks - 1)];
      const float gx = fmaxf(0.f, fmaxf(lox - sx, sx - hix));
      const float gy = fmaxf(0.f, fmaxf(loy - sy, sy - hiy));
      const float gz = fmaxf(0.f, fmaxf(loz - sz, sz - hiz));
      const float lb = d6_sqdist(gx, gy, gz);
      u64 need = __ballot(lane < np && (r == 1 || !(lb >= cmax)));
      if (need != 0ull) {
        while (need & (need - 1ull)) {
          const int i = __builtin_ctzll(need);
          need &= need - 1ull;
          sq_apply<SG>(d6_readlane_f(sx, i), d6_readlane_f(sy, i), d6_readlane_f(sz, i), px, py, pz, pt);
        }
        const int i = __builtin_ctzll(need);
        cmax = sq_rescan_w<SG, kCand>(d6_readlane_f(sx, i), d6_readlane_f(sy, i), d6_readlane_f(sz, i), log2s, px, py, pz, pw, pt, korig_w,
                                      rec, recw, wave, np <= depth_add ? min(kCand, 2) : kCand);
      }
    }
    if (r >= m) break;
    __syncthreads();                                      // records complete
    // ---- B. wave 0: as many picks as the records allow
    if (wave == 0) {
      constexpr int LOG2K = kCand == 2 ? 1 : 2;
      const bool live = lane < kWaves * kCand;
      const int region = live ? lane >> LOG2K : 0;
      const int nc = rec.nc[region];
      const int slot = lane & (kCand - 1);
      const int o = region * kCandMax + slot;
      const bool alive = live && slot < nc;
      float cv = alive ? recw.t[o] : -1.0f;               // current MIN-DISTANCE of this candidate (-1: no candidate in this slot)
      const float cw = alive ? recw.w[o] : 1.0f;          // its weight: score = cv * cw (an empty slot stays at -1)
      const float sv = alive ? rec.v[o] : -1.0f;          // its score as the record was written
      const float qx = rec.x[o], qy = rec.y[o], qz = rec.z[o];
      const int kidx = rec.k[o];
      const bool is_last = live && slot == nc - 1;
      const float bound_v = sq_group_max<kCand>(is_last ? sv : -2.0f);   // SCORE of the record's last candidate (group-uniform)
      const unsigned ub = ((__builtin_bit_cast(unsigned, bound_v) << 1) | 1u) + 2u;
      const unsigned ntk = ~sq_tie_key(kidx, log2s);                                  // larger = earlier in the order
      const unsigned ntk_last = sq_group_max_u32<kCand>(is_last ? ntk : 0u);
      const unsigned thr = !alive ? 0xFFFFFFFFu : ntk >= ntk_last ? ub - 1u : ub;
      const unsigned alt = is_last ? ub : 0u;
      int j = 0;
      const int jmax = min(max_picks, m - r);
      int mwl = 0;
      {
        bool go;
        do {
          const unsigned ekey = (__builtin_bit_cast(unsigned, cv * cw) << 1) + 2u;
          const unsigned key = ekey >= thr ? ekey : alt;
          const unsigned best = sq_wave_max_u32(key);
          const u64 tie = __ballot(key == best);
          int wl = __builtin_ctzll(tie);
          if (__popcll(tie) != 1) wl = sq_min_key_lane(tie, ~ntk);
          const float sx = d6_readlane_f(qx, wl), sy = d6_readlane_f(qy, wl), sz = d6_readlane_f(qz, wl);
          mwl = lane == j ? wl : mwl;
          cv = d6_vmin(cv, d6_sqdist(qx - sx, qy - sy, qz - sz));
          go = (best & 1u) == 0u;
          j += go ? 1 : 0;
        } while (go && j < jmax);
      }
      {
        const float mx = __shfl(qx, mwl), my = __shfl(qy, mwl), mz = __shfl(qz, mwl);
        const int mk = __shfl(kidx, mwl);
        if (lane < j) {
          pick_x[lane] = mx; pick_y[lane] = my; pick_z[lane] = mz;
          idxs[r + lane] = mk + idx_add;
        }
      }
      if (lane == 0) pick_n = j;
    }
    __syncthreads();                                      // picks published
    r += pick_n;
  }
}

}  // namespace

#ifdef DET6D_EXPERIMENTS
extern "C" __attribute__((visibility("default"))) int det6d_dbg_fps_seq_stats(unsigned long long *out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(d6_fps_seq_stats), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}

// Test hook (DET6D_DBG_POISON_LDS=<pattern>, experiments build): the sampler launchers call this right before their sampling
// kernel.  It fills the LDS of every CU with the pattern (one 160 KB workgroup per CU, several waves of them), so that a
// kernel that reads LDS it has not written itself reads the pattern instead of whatever the kernels before left there — in
// a fresh process mostly zeros, which hides such a read (round 4: fps_coop.hip's record slots).
namespace {
__device__ unsigned d6_poison_sink;
__global__ __launch_bounds__(1024) void poison_lds_kernel(unsigned pattern, unsigned words) {
  extern __shared__ unsigned lds[];
  for (unsigned i = threadIdx.x; i < words; i += 1024) lds[i] = pattern;
  __syncthreads();
  if (lds[(threadIdx.x * 97u) % words] != pattern) d6_poison_sink = 1u;     // (keeps the stores)
}
}  // namespace
void det6d_dbg_poison_lds_hook(hipStream_t stream) {
  static const char *env = getenv("DET6D_DBG_POISON_LDS");
  if (!env) return;
  static const unsigned pattern = (unsigned)strtoul(env, nullptr, 0);
  const unsigned bytes = 160u * 1024u;
  DET6D_MAX_DYNAMIC_LDS(poison_lds_kernel, bytes);
  hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(1024), bytes, stream, pattern, bytes / 4u);
  const hipError_t rc = hipGetLastError();
  if (rc != hipSuccess) fprintf(stderr, "det6d_dbg_poison_lds_hook: %s\n", hipGetErrorString(rc));
}
// what an LDS word a workgroup has not written holds, on the CU the probing workgroup lands on (tests of the hook itself)
namespace {
__global__ __launch_bounds__(1024) void probe_lds_kernel(unsigned *out) {
  __shared__ unsigned probe[8192];
  unsigned v = probe[threadIdx.x * 8];
  asm volatile("" : "+v"(v));
  out[blockIdx.x * 1024 + threadIdx.x] = v;
}
}  // namespace
extern "C" __attribute__((visibility("default"))) int det6d_dbg_probe_lds(unsigned *out, int blocks, void *stream) {
  det6d_dbg_poison_lds_hook((hipStream_t)stream);
  hipLaunchKernelGGL(probe_lds_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, out);
  return det6d_check_launch("det6d_dbg_probe_lds");
}
#endif

#ifdef DET6D_EXPERIMENTS
// What-if hook (scripts/r05/whatif_occupancy.py, never a result path): `blocks` workgroups of 16 waves that HOLD R registers per
// lane and `lds` bytes of LDS for `usec` microseconds and issue nothing but s_sleep — the footprint of a sampler workgroup
// without its instructions: what do 32 such workgroups per pass cost the pipeline, by footprint?
namespace {
template <int R>
__global__ __launch_bounds__(1024) void occupy_kernel(int usec, float *sink) {
  extern __shared__ float occ_lds[];
  float r[R];
#pragma unroll
  for (int i = 0; i < R; ++i) r[i] = (float)(threadIdx.x + i);
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)usec * 100ull) {
    __builtin_amdgcn_s_sleep(100);
#pragma unroll
    for (int i = 0; i < R; ++i) asm volatile("" : "+v"(r[i]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < R; ++i) s += r[i];
  if (s == -1.f) { occ_lds[threadIdx.x] = s; sink[0] = occ_lds[0]; }
}
}  // namespace
extern "C" __attribute__((visibility("default"))) int det6d_dbg_occupy(int blocks, int regs, int lds, int usec, float *sink, void *stream) {
  hipStream_t s = (hipStream_t)stream;
#define D6_OCC(R)                                                         \
  do {                                                                    \
    DET6D_MAX_DYNAMIC_LDS(occupy_kernel<R>, 160 * 1024);                  \
    hipLaunchKernelGGL(occupy_kernel<R>, dim3(blocks), dim3(1024), lds, s, usec, sink); \
  } while (0)
  if (regs >= 96) D6_OCC(96); else if (regs >= 80) D6_OCC(80); else if (regs >= 56) D6_OCC(56); else if (regs >= 40) D6_OCC(40); else D6_OCC(16);
#undef D6_OCC
  return det6d_check_launch("det6d_dbg_occupy");
}
#endif

// Called by fps_cells.hip's launcher after the Morton sort and the lane-group ordering (groups of 16 positions).
int det6d_fps_seq_launch(int b, int n, int m, int log2s, long long xyz_bstride, long long idx_bstride, int idx_add,
                         const float *xyz, const int *perm, int *idx, hipStream_t stream) {
  if (n != 16384 && n != 4096) return DET6D_EINVAL;
  static const int max_picks_env = det6d_env_int("DET6D_FPS_SEQ_PICKS", kMaxPicks);
  static const int cands = det6d_env_int("DET6D_FPS_SEQ_CANDS", 4);
  static const int depth_add = det6d_env_int("DET6D_FPS_SEQ_DEPTH_ADD", 1);    // rounds of <= this many picks are followed by short lists
  const int max_picks = max_picks_env < 1 ? 1 : max_picks_env > kMaxPicks ? kMaxPicks : max_picks_env;
#ifdef DET6D_EXPERIMENTS
  det6d_dbg_poison_lds_hook(stream);
#endif
  if (n == 4096)
    hipLaunchKernelGGL((fps_seq_kernel<4, 4>), dim3(b), dim3(1024), 0, stream, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm, idx, max_picks, depth_add);
  else if (cands == 2)
    hipLaunchKernelGGL((fps_seq_kernel<2>), dim3(b), dim3(1024), 0, stream, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm, idx, max_picks, depth_add);
  else
    hipLaunchKernelGGL((fps_seq_kernel<4>), dim3(b), dim3(1024), 0, stream, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm, idx, max_picks, depth_add);
  return det6d_check_launch("det6d_fps (multi-pick)");
}

// The score-weighted form (S-FPS of 16384- / 4096-point clouds): `flags` = the scenes' scratch (b x n ints: the permutation the
// pre-pass wrote; word 0 of a scene becomes its "needs the exact-double sampler" flag), weights / raw scores (b x w_bstride).
int det6d_fps_seq_w_launch(int b, int n, int m, int log2s, long long xyz_bstride, long long idx_bstride, int idx_add,
                           const float *xyz, const int *perm, int *idx, const float *weights, long long w_bstride, float gamma,
                           int w_is_score, int *flags, hipStream_t stream) {
  if (n != 16384 && n != 4096) return DET6D_EINVAL;
  static const int max_picks_env = det6d_env_int("DET6D_FPS_SEQ_PICKS", kMaxPicks);
  static const int depth_add = det6d_env_int("DET6D_FPS_SEQ_DEPTH_ADD", 1);
  const int max_picks = max_picks_env < 1 ? 1 : max_picks_env > kMaxPicks ? kMaxPicks : max_picks_env;
#ifdef DET6D_EXPERIMENTS
  det6d_dbg_poison_lds_hook(stream);
#endif
  if (n == 4096)
    hipLaunchKernelGGL((fps_seq_w_kernel<4, 4>), dim3(b), dim3(1024), 0, stream, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm,
                       idx, max_picks, depth_add, weights, w_bstride, gamma, w_is_score, flags);
  else
    hipLaunchKernelGGL((fps_seq_w_kernel<4, 16>), dim3(b), dim3(1024), 0, stream, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm,
                       idx, max_picks, depth_add, weights, w_bstride, gamma, w_is_score, flags);
  return det6d_check_launch("det6d_fps (multi-pick, score-weighted)");
}
