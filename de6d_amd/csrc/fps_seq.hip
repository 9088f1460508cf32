// fps_seq.hip — look-ahead farthest point sampling (D-FPS) for gfx950: bit for bit the picks of
// farthest_point_sampling_kernel (core/pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:101-222, tie order of its
// shared-memory tree :94-99,159-216), without a barrier and without a rescan on the round's critical path.
//
// The wave-skip sampler (fps_cells.hip) already touches only the waves whose bounding box is near the new sample, but every
// round still pays  rescan -> LDS slot -> s_barrier -> block arg-max  in series: 0.86 us x 4095 rounds.  Here the rounds are
// decided by ONE wave (the sequencer) from small published records; the waves that own the points (owners) apply the picks
// behind it at their own pace.  tests/models/fps_lookahead.py is an executable model of the protocol (random schedules,
// ties, duplicates) and tests/test_fps_lookahead_model.py its soundness check.
//
// Roles.  A workgroup of 16 waves per scene.  Wave 0 is the sequencer (its state beside 16 points per lane does not fit the
// 128 registers a wave of a 1024-thread workgroup may have, so it owns no points; it runs at raised priority: the three owner
// waves on its SIMD take the issue slots it leaves).  The other 15 waves are the owners: 18 points per lane (18 x 64 x 15 =
// 17280 slots for 16384 points; the spare slots at the end of the sorted order hold a min-distance of -1 and never win).
// Regions.  The scene is Morton-sorted (fps_cells.hip: cell_sort_kernel); an owner holds G regions of 64 * SG consecutive
// sorted points (SG = 18 / G slots per lane, ordered by tie key inside the lane), each with a bounding box and its exact
// min-distances in registers.
// Record of a region, published by its owner after every rescan:  tag a = number of picks applied, and the region's top
// (up to) kCand points c_1 > c_2 > .. in the reference's order (value descending, then tie key ascending) as
// {value, index, x, y, z}.  Every other point of the region is ordered AFTER the last candidate, and min-distances only
// decrease.
// Sequencer state per region (lane j = region j): the accepted record plus the CURRENT value cv_i of every candidate, kept
// exact by applying each pick it makes with the scan's own distance expression.  At decision r:
//     X = best of (cv_i, key_i);   the region's maximum is exactly X  iff  X >= (v_last, key_last) of the record;
//     otherwise its maximum is unknown but <= v_last.
// Pick r = the best exact X, provided every unknown region's bound is strictly below its value; otherwise the sequencer polls
// the records (the owners republish after the rescans the blocking picks force on them: liveness).  A rescan is on the
// critical path only when a region loses all its candidates between two of its own republications.
// Owner.  Reads the picks from a ring in LDS, 64 at a time: lane i tests pick r+i against the region boxes (a pick at least
// sqrt(current maximum) away from the box changes nothing: the floating-point box distance is a lower bound of every distance
// the scan would compute, fps_cells.hip), skips the leading run in one step, rescans for the first pick that can matter.
// All communication is 64-bit words {payload, tag} written and read with relaxed atomics: no fence, no barrier after start-up.
#include "common.h"

#include <type_traits>

namespace {

typedef unsigned long long u64;
typedef float sq_f32x2 __attribute__((ext_vector_type(2)));

constexpr int kCand = 4;                 // candidates per record
constexpr int kRecWords = 5 * kCand;     // {value, index, x, y, z} per candidate
constexpr int kRecStride = kRecWords + 1;
constexpr int kRing = 512;               // picks kept in LDS (x, y, z words)
constexpr int kOwners = 15;              // owner waves: 1 .. 15
constexpr int kSlots = 18;               // points per owner lane
constexpr int kOwnerPoints = 64 * kSlots;

__device__ __forceinline__ unsigned sq_bitrev_bits(unsigned v, int bits) {
  return bits == 0 ? 0u : (__builtin_bitreverse32(v) >> (32 - bits));
}
// order key of point k under the reference's tie rule (smaller wins): (bitrev_{log2 S}(k mod S), k)
__device__ __forceinline__ unsigned sq_tie_key(int k, int log2s) {
  return (sq_bitrev_bits((unsigned)k & ((1u << log2s) - 1u), log2s) << (32 - log2s)) | ((unsigned)k >> log2s);
}
__device__ __forceinline__ int sq_key_index(unsigned key, int log2s) {
  const unsigned low = sq_bitrev_bits(key >> (32 - log2s), log2s);
  const unsigned high = key & ((1u << (32 - log2s)) - 1u);
  return (int)((high << log2s) | low);
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

__device__ __forceinline__ u64 sq_pack(unsigned payload, unsigned tag) { return ((u64)tag << 32) | payload; }
__device__ __forceinline__ u64 sq_ld(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void sq_st(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned sq_fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float sq_bitsf(unsigned v) { return __builtin_bit_cast(float, v); }

template <int LO, int HI, int N>
__device__ __forceinline__ void sq_pick(int ws, int wl, const float (&px)[N], const float (&py)[N], const float (&pz)[N],
                                        float &sx, float &sy, float &sz) {
  if constexpr (HI - LO == 1) {
    sx = d6_readlane_f(px[LO], wl);
    sy = d6_readlane_f(py[LO], wl);
    sz = d6_readlane_f(pz[LO], wl);
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (ws < MID) sq_pick<LO, MID>(ws, wl, px, py, pz, sx, sy, sz);
    else sq_pick<MID, HI>(ws, wl, px, py, pz, sx, sy, sz);
  }
}

// one lane of `pay` <- a wave-uniform value (v_writelane_b32 with the lane in m0: the value takes the one constant-bus read)
__device__ __forceinline__ void sq_writelane(int &pay, int uniform_val, int uniform_lane) {
  const int sv = __builtin_amdgcn_readfirstlane(uniform_val), sl = __builtin_amdgcn_readfirstlane(uniform_lane);
  int keep;
  asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
               : "+v"(pay), "=&s"(keep) : "s"(sv), "s"(sl));
}

// Apply the pick (cx, cy, cz) to region g_ (wave-uniform, runtime) of this wave — slots [g*SG, (g+1)*SG) of every lane —,
// extract its record and publish it with tag `tag`.  Per lane the best value (+ slot) and the second best are tracked in the
// scan (med3); the candidates are taken one by one as the best lane head under the order.  A lane knows only its two best
// points, so the list ends with the first candidate that is a lane's SECOND (what is left in that lane is ordered after it, but
// not necessarily after later heads): the record carries its length nc (bits 16.. of the first candidate's index word), the
// reader repeats the last candidate, which keeps "every other point is ordered after the last candidate" true.
template <int G, int SG, int N>
__device__ __forceinline__ float sq_rescan(int g_, float cx, float cy, float cz, int log2s, const float (&px)[N],
                                           const float (&py)[N], const float (&pz)[N], float (&pt)[N],
                                           const unsigned short *korig_g, u64 *rec, int tag) {
  const int lane = threadIdx.x & 63;
  float best = -2.0f, sec = -2.0f;        // below the spare slots' -1
  int bs = 0;
  const sq_f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
  auto visit = [&](int j, int s, float d) {
    const float t = d6_vmin(d, pt[s]);
    pt[s] = t;
    sec = __builtin_amdgcn_fmed3f(best, sec, t);          // second best so far (uses the OLD best)
    const bool up = t > best;
    bs = up ? j : bs;
    best = d6_vmax(best, t);
  };
  auto body = [&](auto gc) {
    constexpr int g = decltype(gc)::value;
#pragma unroll
    for (int q = 0; q < SG / 2; ++q) {
      const int s0 = g * SG + 2 * q;
      const sq_f32x2 dx = sq_f32x2{px[s0], px[s0 + 1]} - c2x;
      const sq_f32x2 dy = sq_f32x2{py[s0], py[s0 + 1]} - c2y;
      const sq_f32x2 dz = sq_f32x2{pz[s0], pz[s0 + 1]} - c2z;
      sq_f32x2 d = dy * dy;
      d = __builtin_elementwise_fma(dx, dx, d);
      d = __builtin_elementwise_fma(dz, dz, d);
      visit(2 * q, s0, d[0]);
      visit(2 * q + 1, s0 + 1, d[1]);
      __builtin_amdgcn_sched_barrier(0);                   // one pair at a time: 88 of the 128 registers hold the points
    }
    if constexpr (SG % 2 == 1) {
      constexpr int s = g * SG + SG - 1;
      visit(SG - 1, s, d6_sqdist(px[s] - cx, py[s] - cy, pz[s] - cz));
    }
  };
  auto second_slot = [&](auto gc) -> int {              // lowest slot != bs holding the lane's second value
    constexpr int g = decltype(gc)::value;
    int ss = 0;
#pragma unroll
    for (int j = SG - 1; j >= 0; --j) ss = (pt[g * SG + j] == sec && j != bs) ? j : ss;
    return ss;
  };
  auto pick = [&](auto gc, int ws, int wl, float &sx, float &sy, float &sz) {
    constexpr int g = decltype(gc)::value;
    sq_pick<g * SG, (g + 1) * SG>(g * SG + ws, wl, px, py, pz, sx, sy, sz);
  };
  auto dispatch = [&](auto &&f) {
    if constexpr (G == 1) f(std::integral_constant<int, 0>{});
    else if constexpr (G == 2) { if (g_ == 0) f(std::integral_constant<int, 0>{}); else f(std::integral_constant<int, 1>{}); }
    else {
      if (g_ == 0) f(std::integral_constant<int, 0>{});
      else if (g_ == 1) f(std::integral_constant<int, 1>{});
      else f(std::integral_constant<int, 2>{});
    }
  };
  static_assert(G >= 1 && G <= 3, "regions per owner");
  dispatch(body);

  int taken = 0;            // this lane's best has been taken
  float head = best;
  float cmax = 0.f;
  int nc = 0, k0 = 0, pay = 0;
#pragma nounroll
  for (int i = 0; i < kCand; ++i) {
    const float wm = d6_wave_max(head);
    if (i == 0) cmax = wm;
    const u64 tie = __ballot(head == wm);
    int wl = __builtin_ctzll(tie);
    int ss = 0;
    bool have_ss = false;
    if (__popcll(tie) != 1) {                              // equal heads: the reference's key decides
      if (__ballot(taken != 0 && head == wm) != 0ull) { dispatch([&](auto gc) { ss = second_slot(gc); }); have_ss = true; }
      const int hs = taken ? ss : bs;
      wl = sq_min_key_lane(tie, sq_tie_key((int)korig_g[lane * SG + hs], log2s));
    }
    const int wtk = d6_readlane_i(taken, wl);
    int ws;
    if (wtk == 0) {
      ws = d6_readlane_i(bs, wl);
    } else {
      if (!have_ss) dispatch([&](auto gc) { ss = second_slot(gc); });
      ws = d6_readlane_i(ss, wl);
    }
    const int k = __builtin_amdgcn_readfirstlane((int)korig_g[wl * SG + ws]);
    float x, y, z;
    dispatch([&](auto gc) { pick(gc, ws, wl, x, y, z); });
    if (i == 0) k0 = k;
    sq_writelane(pay, (int)sq_fbits(wm), 5 * i + 0);
    sq_writelane(pay, k, 5 * i + 1);
    sq_writelane(pay, (int)sq_fbits(x), 5 * i + 2);
    sq_writelane(pay, (int)sq_fbits(y), 5 * i + 3);
    sq_writelane(pay, (int)sq_fbits(z), 5 * i + 4);
    nc = i + 1;
    if (lane == wl) { taken = 1; head = sec; }
    if (wtk != 0) break;                                   // a lane is exhausted: the list ends here
  }
  sq_writelane(pay, k0 | (nc << 16), 1);
  if (lane < 5 * nc) sq_st(rec + lane, sq_pack((unsigned)pay, (unsigned)tag));
  return cmax;
}

__device__ int d6_fps_seq_timeouts;

#ifdef DET6D_EXPERIMENTS
// scripts/experiments only: protocol counters of workgroup 0 of the last launch (det6d_dbg_fps_seq_stats)
__device__ unsigned long long d6_fps_seq_stats[16];
#define SQ_STAT(i, v) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicAdd(&sq_stats_lds[i], (unsigned long long)(v)); } while (0)
#define SQ_CLK() clock64()
#else
#define SQ_STAT(i, v) do { } while (0)
#define SQ_CLK() 0ll
#endif

// a wave that makes no progress for this many iterations leaves (see the watchdog notes in the sequencer)
constexpr int kIdleLimit = 1 << 21;

template <int G>
__device__ __forceinline__ void sq_owner(int ow, int n, int m, int log2s, const float *__restrict__ xyz, const int *__restrict__ perm,
                                         unsigned short *korig, u64 *rec, const u64 *hist, int *progress,
                                         unsigned long long *sq_stats_lds) {
  constexpr int SG = kSlots / G;
  static_assert(SG * G == kSlots, "region layout");
  const int lane = threadIdx.x & 63;
  float px[kSlots], py[kSlots], pz[kSlots], pt[kSlots];
  float lox[G], loy[G], loz[G], hix[G], hiy[G], hiz[G], cmax[G];
  {
    // this lane's sorted positions, ordered by tie key inside every region (spare slots: key 0xFFFFFFFF, last)
    int kk[kSlots];
    unsigned key[kSlots];
    const int klast = perm[n - 1];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < SG; ++j) {
        const int pos = ((ow * G + g) * 64 + lane) * SG + j;
        const bool real = pos < n;
        kk[g * SG + j] = real ? perm[pos] : klast;
        key[g * SG + j] = real ? sq_tie_key(kk[g * SG + j], log2s) : 0xFFFFFFFFu;
      }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int pass = 0; pass < SG; ++pass)                 // odd-even transposition sort of SG entries
#pragma unroll
        for (int j = pass & 1; j + 1 < SG; j += 2) {
          const int a = g * SG + j, b = a + 1;
          const bool sw = key[b] < key[a];
          const unsigned ka = key[a], kb = key[b];
          const int ia = kk[a], ib = kk[b];
          key[a] = sw ? kb : ka; key[b] = sw ? ka : kb;
          kk[a] = sw ? ib : ia; kk[b] = sw ? ia : ib;
        }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float ax = 3.0e38f, ay = 3.0e38f, az = 3.0e38f, bx = -3.0e38f, by = -3.0e38f, bz = -3.0e38f;
#pragma unroll
      for (int j = 0; j < SG; ++j) {
        const int s = g * SG + j;
        const int k = kk[s];
        korig[((ow * G + g) * 64 + lane) * SG + j] = (unsigned short)k;
        px[s] = xyz[(size_t)k * 3 + 0];
        py[s] = xyz[(size_t)k * 3 + 1];
        pz[s] = xyz[(size_t)k * 3 + 2];
        asm volatile("" : "+v"(px[s]), "+v"(py[s]), "+v"(pz[s]));
        pt[s] = key[s] == 0xFFFFFFFFu ? -1.0f : 1e10f;     // a spare slot never exceeds a real point (min-distances >= 0)
        ax = d6_vmin(ax, px[s]); bx = d6_vmax(bx, px[s]);
        ay = d6_vmin(ay, py[s]); by = d6_vmax(by, py[s]);
        az = d6_vmin(az, pz[s]); bz = d6_vmax(bz, pz[s]);
      }
      lox[g] = d6_wave_min(ax); hix[g] = d6_wave_max(bx);
      loy[g] = d6_wave_min(ay); hiy[g] = d6_wave_max(by);
      loz[g] = d6_wave_min(az); hiz[g] = d6_wave_max(bz);
      cmax[g] = __builtin_inff();
    }
  }

  int r_next = 0;                                         // next pick to apply = number of picks applied
  int idle = 0;
  while (r_next < m) {
    if (++idle > kIdleLimit) return;                      // watchdog (see the sequencer)
    // the next 64 picks at once
    const int e = r_next + lane;
    const u64 *hp = hist + (e & (kRing - 1)) * 3;
    const u64 wx = sq_ld(hp), wy = sq_ld(hp + 1), wz = sq_ld(hp + 2);
    const unsigned et = (unsigned)e + 1u;                 // ring tags are pick index + 1 (0 = never written)
    const bool valid = e < m && (unsigned)(wx >> 32) == et && (unsigned)(wy >> 32) == et && (unsigned)(wz >> 32) == et;
    const u64 vmask = __ballot(valid);
    const int nvalid = ~vmask == 0ull ? 64 : __builtin_ctzll(~vmask);
    if (nvalid == 0) {
      if (ow == 4) SQ_STAT(12, 1);
      __builtin_amdgcn_s_sleep(4);
      continue;
    }
    if (ow == 4) SQ_STAT(14, 1);
    idle = 0;
    const float sx = sq_bitsf((unsigned)wx), sy = sq_bitsf((unsigned)wy), sz = sq_bitsf((unsigned)wz);
    u64 need[G];
    u64 any = 0ull;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float gx = fmaxf(0.f, fmaxf(lox[g] - sx, sx - hix[g]));
      const float gy = fmaxf(0.f, fmaxf(loy[g] - sy, sy - hiy[g]));
      const float gz = fmaxf(0.f, fmaxf(loz[g] - sz, sz - hiz[g]));
      const float lb = d6_sqdist(gx, gy, gz);
      need[g] = __ballot(!(lb >= cmax[g]));
      any |= need[g];
    }
    const int first = any == 0ull ? 64 : __builtin_ctzll(any);
    const int nskip = first < nvalid ? first : nvalid;
    r_next += nskip;
    if (nskip < nvalid) {                                   // pick r_next can change this wave's points
      const float cx = d6_readlane_f(sx, nskip), cy = d6_readlane_f(sy, nskip), cz = d6_readlane_f(sz, nskip);
      r_next += 1;
      unsigned needbits = 0u;
#pragma unroll
      for (int g = 0; g < G; ++g) needbits |= (unsigned)((need[g] >> nskip) & 1ull) << g;
#pragma unroll
      for (int g = 0; g < G; ++g) {                         // (a runtime g — one copy of the rescan — makes the allocator shuffle and spill the point registers)
        if ((needbits >> g) & 1u) {
          if (ow == 4) SQ_STAT(13, 1);
          SQ_STAT(7, 1);
          const float nm = sq_rescan<G, SG>(g, cx, cy, cz, log2s, px, py, pz, pt, korig + (size_t)(ow * G + g) * 64 * SG,
                                            rec + (ow * G + g) * kRecStride, r_next);
#pragma unroll
          for (int gg = 0; gg < G; ++gg) cmax[gg] = gg == g ? nm : cmax[gg];
        }
      }
    }
    if (lane == 0) __hip_atomic_store(&progress[ow], r_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

template <int G>
__device__ __forceinline__ void sq_sequencer(int m, int log2s, int idx_add, int *__restrict__ idxs, const u64 *rec, u64 *hist,
                                             const int *progress, unsigned long long *sq_stats_lds) {
  constexpr int NR = kOwners * G;
  static_assert(NR <= 64, "one region per sequencer lane");
  const int lane = threadIdx.x & 63;
  const bool live = lane < NR;
  const u64 *rp = rec + (live ? lane : 0) * kRecStride;
  int r_dec = 1;                                          // next pick to decide
  int minprog = 0;                                        // lower bound of every owner's progress
  unsigned a_tag = 0;
  float cv[kCand], qx[kCand], qy[kCand], qz[kCand];
  unsigned nkey[kCand];                                   // ~tie key: larger = better
  float bound_v = __builtin_inff();
#pragma unroll
  for (int i = 0; i < kCand; ++i) { cv[i] = 0.f; qx[i] = qy[i] = qz[i] = 0.f; nkey[i] = 0u; }
  int idle = 0;
#ifdef DET6D_EXPERIMENTS
  const long long t_begin = clock64();
#endif
  while (r_dec < m) {
    if (++idle > kIdleLimit) {
      // Watchdog: the protocol has no way to stall (see the header), but a sequencer that decides nothing for ~2^21
      // iterations (seconds) completes the index list with in-range placeholders and counts the event
      // (det6d_fps_seq_timeouts) instead of hanging the device; the owners leave by their own counters.
      for (int i = r_dec + lane; i < m; i += 64) idxs[i] = idx_add;
      if (lane == 0) atomicAdd(&d6_fps_seq_timeouts, 1);
      return;
    }
    SQ_STAT(0, 1);
    const long long tq0 = SQ_CLK();
    // -- 1. new records?
    const u64 w0 = sq_ld(rp);
    const unsigned t0 = (unsigned)(w0 >> 32);
    const bool changed = live && t0 > a_tag;
    if (__ballot(changed) != 0ull) {
      SQ_STAT(4, 1);
      // candidate by candidate (a whole record in registers would be 40 VGPRs); a torn record (the owner is rewriting it)
      // leaves the region WITHOUT a record (unknown, unbounded) until the next poll: sound, and rare
      bool ok = changed;
      int nc = 1;
#pragma unroll
      for (int i = 0; i < kCand; ++i) {
        const u64 wv = sq_ld(rp + 5 * i), wk = sq_ld(rp + 5 * i + 1), wx = sq_ld(rp + 5 * i + 2), wy = sq_ld(rp + 5 * i + 3),
                  wz = sq_ld(rp + 5 * i + 4);
        if (i == 0) nc = (int)((unsigned)wk >> 16);
        const bool used = i == 0 || i < nc;
        ok = ok && (!used || ((unsigned)(wv >> 32) == t0 && (unsigned)(wk >> 32) == t0 && (unsigned)(wx >> 32) == t0 &&
                              (unsigned)(wy >> 32) == t0 && (unsigned)(wz >> 32) == t0));
        if (changed) {
          if (used) {
            cv[i] = sq_bitsf((unsigned)wv);
            nkey[i] = ~sq_tie_key((int)((unsigned)wk & 0xFFFFu), log2s);
            qx[i] = sq_bitsf((unsigned)wx);
            qy[i] = sq_bitsf((unsigned)wy);
            qz[i] = sq_bitsf((unsigned)wz);
          } else if (i > 0) {
            cv[i] = cv[i - 1]; nkey[i] = nkey[i - 1]; qx[i] = qx[i - 1]; qy[i] = qy[i - 1]; qz[i] = qz[i - 1];
          }
        }
      }
      if (changed) {
        bound_v = ok ? cv[kCand - 1] : __builtin_inff();
        a_tag = ok ? t0 : 0u;
      }
      if (__ballot(ok) != 0ull) {
        // picks made since those records: t0 .. r_dec-1
        const int from = (int)d6_wave_min(ok ? (float)t0 : 3.0e38f);
        SQ_STAT(5, r_dec - from);
        SQ_STAT(6, __popcll(__ballot(ok)));
        for (int i = from; i < r_dec; ++i) {
          const u64 *hp = hist + (i & (kRing - 1)) * 3;
          const float sx = sq_bitsf((unsigned)sq_ld(hp)), sy = sq_bitsf((unsigned)sq_ld(hp + 1)), sz = sq_bitsf((unsigned)sq_ld(hp + 2));
          if (ok && (int)t0 <= i) {
#pragma unroll
            for (int c = 0; c < kCand; ++c) cv[c] = d6_vmin(cv[c], d6_sqdist(qx[c] - sx, qy[c] - sy, qz[c] - sz));
          }
        }
      }
    }
    const long long tq1 = SQ_CLK();
    SQ_STAT(8, tq1 - tq0);
    // -- 2. decide
    if (r_dec - minprog >= kRing - 64) {                   // the ring slot about to be overwritten may still be unread
      const float p = lane < kOwners ? (float)__hip_atomic_load(&progress[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 3.0e38f;
      minprog = (int)d6_wave_min(p);
      if (r_dec - minprog >= kRing - 64) { SQ_STAT(3, 1); __builtin_amdgcn_s_sleep(1); continue; }
    }
    // (value bits, ~key) as ONE signed 64-bit number: min-distances are >= +0 (their float order is their integer order), the
    // -1 of a spare slot is negative, a smaller key wins among equal values
    long long X = (long long)(((u64)sq_fbits(cv[0]) << 32) | nkey[0]);
    int ci = 0;
#pragma unroll
    for (int c = 1; c < kCand; ++c) {
      const long long cc = (long long)(((u64)sq_fbits(cv[c]) << 32) | nkey[c]);
      const bool gt = cc > X;
      X = gt ? cc : X;
      ci = gt ? c : ci;
    }
    const long long B = (long long)(((u64)sq_fbits(bound_v) << 32) | nkey[kCand - 1]);
    const bool exact = live && a_tag != 0u && X >= B;
    const float xv = sq_bitsf((unsigned)((u64)X >> 32));
    const float ev = exact ? xv : -1.0f;
    const float ub = live && !exact ? bound_v : -1.0f;
    const float E = d6_wave_max(ev);
    const float UB = d6_wave_max(ub);
    if (!(UB < E)) { SQ_STAT(2, 1); continue; }            // some region's maximum is unknown and may be the largest: poll
    const u64 tie = __ballot(exact && ev == E);
    int wl = __builtin_ctzll(tie);
    if (__popcll(tie) != 1) wl = sq_min_key_lane(tie, ~(unsigned)X);
    const int cw = d6_readlane_i(ci, wl);
    const int k = sq_key_index(~(unsigned)__builtin_amdgcn_readlane((int)(unsigned)X, wl), log2s);
    float sx, sy, sz;
    if (cw == 0) { sx = d6_readlane_f(qx[0], wl); sy = d6_readlane_f(qy[0], wl); sz = d6_readlane_f(qz[0], wl); }
    else if (cw == 1) { sx = d6_readlane_f(qx[1], wl); sy = d6_readlane_f(qy[1], wl); sz = d6_readlane_f(qz[1], wl); }
    else if (cw == 2) { sx = d6_readlane_f(qx[2], wl); sy = d6_readlane_f(qy[2], wl); sz = d6_readlane_f(qz[2], wl); }
    else { sx = d6_readlane_f(qx[3], wl); sy = d6_readlane_f(qy[3], wl); sz = d6_readlane_f(qz[3], wl); }
    if (lane < 3) sq_st(hist + (r_dec & (kRing - 1)) * 3 + lane, sq_pack(sq_fbits(lane == 0 ? sx : lane == 1 ? sy : sz), (unsigned)r_dec + 1u));
    if (lane == 0) idxs[r_dec] = k + idx_add;
#pragma unroll
    for (int c = 0; c < kCand; ++c) cv[c] = d6_vmin(cv[c], d6_sqdist(qx[c] - sx, qy[c] - sy, qz[c] - sz));
    ++r_dec;
    idle = 0;
    SQ_STAT(1, 1);
    SQ_STAT(9, SQ_CLK() - tq1);
  }
#ifdef DET6D_EXPERIMENTS
  SQ_STAT(11, clock64() - t_begin);
  __builtin_amdgcn_s_waitcnt(0);
  if (blockIdx.x == 0 && lane < 16) d6_fps_seq_stats[lane] = sq_stats_lds[lane];
#endif
}

// One workgroup of 16 waves per scene; G regions per owner wave.  `perm`: the scene's Morton permutation (n entries).
template <int G>
__global__ __launch_bounds__(1024) void fps_seq_kernel(int n, int m, int log2s, long long xyz_bstride, long long idx_bstride,
                                                       int idx_add, const float *__restrict__ xyz,
                                                       const int *__restrict__ perm, int *__restrict__ idxs) {
  constexpr int NR = kOwners * G;
  __shared__ unsigned short korig[kOwners * kOwnerPoints];   // sorted slot -> original index
  __shared__ u64 rec[NR * kRecStride];
  __shared__ u64 hist[kRing * 3];
  __shared__ int progress[kOwners];
#ifdef DET6D_EXPERIMENTS
  __shared__ unsigned long long sq_stats[16];
  if (threadIdx.x < 16) sq_stats[threadIdx.x] = 0ull;
#else
  unsigned long long *sq_stats = nullptr;
#endif
  const int h = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(h >> 6);
  xyz += (size_t)blockIdx.x * xyz_bstride;
  perm += (size_t)blockIdx.x * n;
  idxs += (size_t)blockIdx.x * idx_bstride;
  for (int w = h; w < kRing * 3; w += 1024) hist[w] = 0ull;
  for (int w = h; w < NR * kRecStride; w += 1024) rec[w] = 0ull;
  if (h < kOwners) progress[h] = 0;
  __syncthreads();                                        // the only barrier: the tags are clear
  // pick 0 is point 0 (sampling_gpu.cu:131-133): ring entry 0; every region's box test passes against its initial
  // maximum (+inf), so every owner starts with a rescan and publishes its first record (tag 1)
  if (h < 3) sq_st(hist + h, sq_pack(sq_fbits(xyz[h]), 1u));
  if (h == 0) idxs[0] = idx_add;
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    sq_sequencer<G>(m, log2s, idx_add, idxs, rec, hist, progress, sq_stats);
    return;
  }
  sq_owner<G>(wave - 1, n, m, log2s, xyz, perm, korig, rec, hist, progress, sq_stats);
}

}  // namespace

#ifdef DET6D_EXPERIMENTS
// 0 sequencer steps, 1 decisions, 2 blocked polls, 3 ring waits, 4 polls that saw a new record, 5 picks replayed onto new
// records, 6 records accepted, 7 rescans (all owners), 8 cycles polling / accepting, 9 cycles deciding, 11 total cycles,
// 12 empty polls / 13 rescans / 14 productive steps of owner 4
extern "C" __attribute__((visibility("default"))) int det6d_dbg_fps_seq_stats(unsigned long long *out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(d6_fps_seq_stats), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

// number of sampler workgroups that ever gave up (watchdog above); 0 in a healthy process.  Synchronises the device.
DET6D_API int det6d_fps_seq_timeouts(void) {
  int v = -1;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(d6_fps_seq_timeouts), sizeof(int)) != hipSuccess) return -1;
  return v;
}

// Called by fps_cells.hip's launcher after the Morton sort (`perm`).
int det6d_fps_seq_launch(int b, int n, int m, int log2s, int regions_per_wave, long long xyz_bstride, long long idx_bstride,
                         int idx_add, const float *xyz, const int *perm, int *idx, hipStream_t stream) {
  if (n != 16384) return DET6D_EINVAL;
  if (regions_per_wave == 1)
    hipLaunchKernelGGL((fps_seq_kernel<1>), dim3(b), dim3(1024), 0, stream, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm, idx);
  else if (regions_per_wave == 2)
    hipLaunchKernelGGL((fps_seq_kernel<2>), dim3(b), dim3(1024), 0, stream, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm, idx);
  else if (regions_per_wave == 3)
    hipLaunchKernelGGL((fps_seq_kernel<3>), dim3(b), dim3(1024), 0, stream, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm, idx);
  else
    return DET6D_EINVAL;
  return det6d_check_launch("det6d_fps (look-ahead)");
}
