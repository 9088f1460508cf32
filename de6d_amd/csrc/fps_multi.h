// fps_multi.h — pieces shared by the multi-pick farthest point samplers (fps_seq.hip: one workgroup per 16384-point scene,
// experiments build; fps_coop.hip: 2 / 4 cooperating workgroups per 32768 / 65536-point scene): the record a wave publishes
// after a rescan (its top K points in the reference's order), the rescan that extracts it, and the cross-lane reductions of
// the sequencer.  The decision rule is described at the top of fps_seq.hip; tests/models/fps_lookahead.py is its executable
// model.
#pragma once
#include "common.h"

namespace {

typedef unsigned long long u64;
typedef float sq_f32x2 __attribute__((ext_vector_type(2)));

constexpr int kCandMax = 4;              // candidates per record = sequencer lanes per region (2 or 4)
constexpr int kWaves = 16, kSlots = 16;  // 16 x 64 x 16 = 16384 points
constexpr int kMaxPicks = 32;            // picks per round at most (one workgroup per scene)
constexpr int kCoopMaxPicks = 64;        // ... of the cooperative form (one per lane of the owners' box test)
static_assert(kWaves * kCandMax == 64, "one sequencer lane per candidate");

__device__ __forceinline__ unsigned sq_bitrev_bits(unsigned v, int bits) {
  return bits == 0 ? 0u : (__builtin_bitreverse32(v) >> (32 - bits));
}
// order key of point k under the reference's tie rule (smaller wins): (bitrev_{log2 S}(k mod S), k)
__device__ __forceinline__ unsigned sq_tie_key(int k, int log2s) {
  return (sq_bitrev_bits((unsigned)k & ((1u << log2s) - 1u), log2s) << (32 - log2s)) | ((unsigned)k >> log2s);
}
// the point a tie key belongs to
__device__ __forceinline__ int sq_tie_key_point(unsigned key, int log2s) {
  if (log2s == 0) return (int)key;
  return (int)(((key & ((1u << (32 - log2s)) - 1u)) << log2s) | sq_bitrev_bits(key >> (32 - log2s), log2s));
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

// slot ws (wave-uniform) of this lane's coordinate registers: scalar binary search down to the statically indexed slot.
// (The empty asm in the leaf keeps the 16 leaves apart: without it the compiler merges them into ONE load with a computed
// index, which moves px / py from registers into scratch memory — every sweep of the kernel then re-reads them from there.)
template <int LO, int HI, int N>
__device__ __forceinline__ void sq_select(int ws, const float (&px)[N], const float (&py)[N], const float (&pz)[N],
                                          float &x, float &y, float &z) {
  if constexpr (HI - LO == 1) {
    x = px[LO]; y = py[LO]; z = pz[LO];
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (ws < MID) sq_select<LO, MID>(ws, px, py, pz, x, y, z);
    else sq_select<MID, HI>(ws, px, py, pz, x, y, z);
  }
}

// the same search over five register arrays (score-weighted sampler: coordinates, min-distance and weight of a slot)
template <int LO, int HI, int N>
__device__ __forceinline__ void sq_select5(int ws, const float (&px)[N], const float (&py)[N], const float (&pz)[N],
                                           const float (&pt)[N], const float (&pw)[N], float &x, float &y, float &z, float &t, float &w) {
  if constexpr (HI - LO == 1) {
    x = px[LO]; y = py[LO]; z = pz[LO]; t = pt[LO]; w = pw[LO];
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(t), "+v"(w));
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (ws < MID) sq_select5<LO, MID>(ws, px, py, pz, pt, pw, x, y, z, t, w);
    else sq_select5<MID, HI>(ws, px, py, pz, pt, pw, x, y, z, t, w);
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
// group maxima (pairs or quads) of N values at once: the DPP steps of the N reductions interleave, no wait states between them
template <int K, int N>
__device__ __forceinline__ void sq_group_max_n(const float (&a)[N], float (&r)[N]) {
  static_assert(N == 1 || N == 2 || N == 4, "values per call");
  if constexpr (N == 4) {
    if constexpr (K == 2) {
      asm volatile("s_nop 1\n\t"
                   "v_max_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\t"
                   : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
    } else {
      asm volatile("s_nop 1\n\t"
                   "v_max_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\t"
                   : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
    }
  } else if constexpr (N == 2) {
    if constexpr (K == 2) {
      asm volatile("s_nop 1\n\t"
                   "v_max_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\t"
                   : "=&v"(r[0]), "=&v"(r[1]) : "v"(a[0]), "v"(a[1]));
    } else {
      sq_quad_max2(a[0], a[1], r[0], r[1]);
    }
  } else {
    if constexpr (K == 2) r[0] = sq_pair_max(a[0]);
    else r[0] = sq_quad_max(a[0]);
  }
}

template <int K>
__device__ __forceinline__ float sq_group_max(float a) {
  if constexpr (K == 2) return sq_pair_max(a);
  else return sq_quad_max(a);
}

// max over the lanes of a group (pair or quad) of an unsigned value, in all its lanes
template <int K>
__device__ __forceinline__ unsigned sq_group_max_u32(unsigned a) {
  unsigned r;
  if constexpr (K == 2)
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 : "=&v"(r) : "v"(a));
  else
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 : "=&v"(r) : "v"(a));
  return r;
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

// score-weighted sampler (round 6): min-distance and weight of every candidate beside its score (SqRecords::v), so that the
// sequencer can keep the SCORE of a candidate exact (score = fp32(t * w): fps.hip, S-FPS in exact fp32)
struct SqRecordsW {
  float t[64], w[64];
};

// Exact duplicates inside a lane.  A lane's slots are in the reference's order, so a point P' with the coordinates of an
// earlier slot P of the same lane has P's min-distance at all times and loses every tie against it: P' is never the arg-max
// of anything.  Its min-distance starts at 0 instead of 1e10, which takes it out of the lane's best / second bookkeeping —
// otherwise a duplicated point costs its region's record two candidates that the same pick kills (clouds padded by
// repetition, data_processor.py:170-175: "every point twice" ran at one pick per round).  The value 0 is what P' holds in
// truth once P or P' has been picked, and a bound too low for P' is harmless: whenever P' is above it, so is P.
template <int SG>
__device__ __forceinline__ void sq_hide_lane_duplicates(const float (&px)[SG], const float (&py)[SG], const float (&pz)[SG],
                                                        float (&pt)[SG]) {
#pragma unroll
  for (int j = 1; j < SG; ++j) {
    bool dup = false;
#pragma unroll
    for (int i = 0; i < j; ++i) dup |= px[i] == px[j] && py[i] == py[j] && pz[i] == pz[j];
    pt[j] = dup ? 0.f : pt[j];
  }
}

// score-weighted form: a duplicate must carry the same weight too (equal coordinates with another weight is another score)
template <int SG>
__device__ __forceinline__ void sq_hide_lane_duplicates_w(const float (&px)[SG], const float (&py)[SG], const float (&pz)[SG],
                                                          const float (&pw)[SG], float (&pt)[SG]) {
#pragma unroll
  for (int j = 1; j < SG; ++j) {
    bool dup = false;
#pragma unroll
    for (int i = 0; i < j; ++i) dup |= px[i] == px[j] && py[i] == py[j] && pz[i] == pz[j] && pw[i] == pw[j];
    pt[j] = dup ? 0.f : pt[j];
  }
}

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
                                           int wave, int depth = kCand) {
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
  // `depth` (1 .. kCand, wave-uniform): how many candidates to list.  A shorter list is as valid as a long one — only less
  // useful to the sequencer — and an extraction step costs ~600 cycles: the callers ask for few while rounds make few picks.
#pragma nounroll
  for (int i = 0; i < depth; ++i) {
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

// sq_rescan of the SCORE-WEIGHTED sampler (S-FPS: arg-max of fp32(min-distance x weight), sampling_gpu.cu:419-540): the same
// scan and extraction on the scores; min-distances stay what the picks are applied to.  The record carries score, index,
// coordinates (SqRecords) and min-distance + weight (SqRecordsW).  Returns the region's maximal MIN-DISTANCE — what the
// owners' box test needs: a pick at least sqrt(that) away from the box lowers no min-distance of the region, hence no score.
template <int SG, int kCand>
__device__ __forceinline__ float sq_rescan_w(float cx, float cy, float cz, int log2s, const float (&px)[SG], const float (&py)[SG],
                                             const float (&pz)[SG], const float (&pw)[SG], float (&pt)[SG],
                                             const unsigned short *korig_w, SqRecords &rec, SqRecordsW &recw, int wave,
                                             int depth = kCand) {
  const int lane = threadIdx.x & 63;
  float best = -1.0f, sec = -1.0f, tm = 0.f;
  int bs = 0;
  const sq_f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
  auto visit = [&](int j, float d) {
    const float t = d6_vmin(d, pt[j]);
    pt[j] = t;
    tm = d6_vmax(tm, t);
    const float sc = t * pw[j];
    sec = __builtin_amdgcn_fmed3f(best, sec, sc);         // second best so far (uses the OLD best)
    const bool up = sc > best;
    bs = up ? j : bs;
    best = d6_vmax(best, sc);
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
  auto second_slot = [&]() -> int {                      // lowest slot != bs holding the lane's second SCORE
    float target = sec;
    int skip = bs;
    asm volatile("" : "+v"(target), "+v"(skip));
    int ss = 0;
#pragma unroll
    for (int j = SG - 1; j >= 0; --j) ss = (pt[j] * pw[j] == target && j != skip) ? j : ss;
    return ss;
  };
  int taken = 0;
  float head = best;
  int nc = 0;
#pragma nounroll
  for (int i = 0; i < depth; ++i) {
    const float wm = d6_wave_max(head);
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
      float x, y, z, t, w;
      sq_select5<0, SG>(ws, px, py, pz, pt, pw, x, y, z, t, w);
      const int o = wave * kCandMax + i;
      rec.v[o] = wm;
      rec.k[o] = (int)korig_w[lane * SG + ws];
      rec.x[o] = x; rec.y[o] = y; rec.z[o] = z;
      recw.t[o] = t; recw.w[o] = w;
      taken = 1;
      head = sec;
    }
    nc = i + 1;
    if (wtk != 0) break;                                   // a lane is exhausted: the list ends here
  }
  if (lane == 0) rec.nc[wave] = nc;
  return d6_wave_max(tm);
}

}  // namespace
