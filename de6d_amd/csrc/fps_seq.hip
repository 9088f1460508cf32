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

constexpr int kCand = 4;                 // candidates per record = sequencer lanes per region
constexpr int kRecWords = 5 * kCand + 2; // {value, index, x, y, z} per candidate, then {count}, {value of the last candidate}
constexpr int kRecStride = kRecWords + 1;
constexpr int kWordCount = 5 * kCand, kWordBound = 5 * kCand + 1;
constexpr int kRing = 512;               // picks kept in LDS (x, y, z words)
constexpr int kOwners = 15;              // owner waves 1 .. 15, one region each
constexpr int kSlots = 18;               // points per owner lane
constexpr int kOwnerPoints = 64 * kSlots;
static_assert(kOwners * kCand <= 64, "one sequencer lane per candidate");

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

__device__ __forceinline__ u64 sq_pack(unsigned payload, unsigned tag) { return ((u64)tag << 32) | payload; }
__device__ __forceinline__ u64 sq_ld(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void sq_st(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned sq_fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float sq_bitsf(unsigned v) { return __builtin_bit_cast(float, v); }

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

// three wave-uniform values into lane `uniform_lane` of three registers (v_writelane_b32 with the lane in m0: the value
// operand takes the one constant-bus read)
__device__ __forceinline__ void sq_writelane3(float &a, float &b, float &c, float va, float vb, float vc, int uniform_lane) {
  const int sa = __builtin_amdgcn_readfirstlane((int)sq_fbits(va)), sb = __builtin_amdgcn_readfirstlane((int)sq_fbits(vb)),
            sc = __builtin_amdgcn_readfirstlane((int)sq_fbits(vc)), sl = __builtin_amdgcn_readfirstlane(uniform_lane);
  int keep;
  asm volatile("s_mov_b32 %3, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\t"
               "v_writelane_b32 %2, %6, m0\n\ts_mov_b32 m0, %3"
               : "+v"(a), "+v"(b), "+v"(c), "=&s"(keep) : "s"(sa), "s"(sb), "s"(sc), "s"(sl));
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

// Apply the pick (cx, cy, cz) to this wave's region, extract its record and publish it with tag `tag`.  Per lane the best
// value (+ slot) and the second best are tracked in the scan (med3); the candidates are taken one by one as the best lane head
// under the order, and the lane that holds a candidate writes it to the record itself.  A lane knows only its two best points,
// so the list ends with the first candidate that is a lane's SECOND (what is left in that lane is ordered after it, but not
// necessarily after later heads).  Words: candidate i at [5i, 5i+5), then the count, then — LAST, it is the word the sequencer
// polls — the value of the last candidate.  Returns the region's maximum.
template <int SG>
__device__ __forceinline__ float sq_rescan(float cx, float cy, float cz, int log2s, const float (&px)[SG], const float (&py)[SG],
                                           const float (&pz)[SG], float (&pt)[SG], const unsigned short *korig_w, u64 *rec, int tag) {
  const int lane = threadIdx.x & 63;
  float best = -2.0f, sec = -2.0f;        // below the spare slots' -1
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
    int ss = 0;
#pragma unroll
    for (int j = SG - 1; j >= 0; --j) ss = (pt[j] == sec && j != bs) ? j : ss;
    return ss;
  };

  int taken = 0;            // this lane's best has been taken
  float head = best;
  float cmax = 0.f, vlast = 0.f;
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
      const unsigned k = korig_w[lane * SG + ws];
      u64 *r = rec + 5 * i;
      sq_st(r + 0, sq_pack(sq_fbits(wm), (unsigned)tag));
      sq_st(r + 1, sq_pack(k, (unsigned)tag));
      sq_st(r + 2, sq_pack(sq_fbits(x), (unsigned)tag));
      sq_st(r + 3, sq_pack(sq_fbits(y), (unsigned)tag));
      sq_st(r + 4, sq_pack(sq_fbits(z), (unsigned)tag));
      taken = 1;
      head = sec;
    }
    nc = i + 1;
    vlast = wm;
    if (wtk != 0) break;                                   // a lane is exhausted: the list ends here
  }
  if (lane == 0) {
    sq_st(rec + kWordCount, sq_pack((unsigned)nc, (unsigned)tag));
    sq_st(rec + kWordBound, sq_pack(sq_fbits(vlast), (unsigned)tag));
  }
  return cmax;
}

// min-distances of this wave's points against one more pick; no arg-max bookkeeping (a later sq_rescan of the same batch does
// it once for all the picks)
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

__device__ int d6_fps_seq_timeouts;

#ifdef DET6D_EXPERIMENTS
// scripts/experiments only: protocol counters of workgroup 0 of the last launch (det6d_dbg_fps_seq_stats)
__device__ unsigned long long d6_fps_seq_stats[16];
#define SQ_STAT(i, v) do { const unsigned long long sv_ = (unsigned long long)(v); if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicAdd(&sq_stats_lds[i], sv_); } while (0)
#define SQ_CLK() clock64()
#else
#define SQ_STAT(i, v) do { } while (0)
#define SQ_CLK() 0ll
#endif

// a wave that makes no progress for this many iterations leaves (see the watchdog notes in the sequencer)
constexpr int kIdleLimit = 1 << 21;

__device__ __forceinline__ void sq_owner(int ow, int n, int m, int log2s, const float *__restrict__ xyz, const int *__restrict__ perm,
                                         unsigned short *korig, u64 *rec, const u64 *hist, int *progress,
                                         unsigned long long *sq_stats_lds, int tune_sleep) {
  constexpr int SG = kSlots;
  const int lane = threadIdx.x & 63;
  unsigned short *korig_w = korig + (size_t)ow * kOwnerPoints;
  rec += ow * kRecStride;
  float px[SG], py[SG], pz[SG], pt[SG];
  float lox, loy, loz, hix, hiy, hiz;
  {
    // this lane's sorted positions, ordered by tie key (spare slots: key 0xFFFFFFFF, last)
    int kk[SG];
    unsigned key[SG];
    const int klast = perm[n - 1];
#pragma unroll
    for (int j = 0; j < SG; ++j) {
      const int pos = (ow * 64 + lane) * SG + j;
      const bool real = pos < n;
      kk[j] = real ? perm[pos] : klast;
      key[j] = real ? sq_tie_key(kk[j], log2s) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int pass = 0; pass < SG; ++pass)                   // odd-even transposition sort of SG entries
#pragma unroll
      for (int j = pass & 1; j + 1 < SG; j += 2) {
        const bool sw = key[j + 1] < key[j];
        const unsigned ka = key[j], kb = key[j + 1];
        const int ia = kk[j], ib = kk[j + 1];
        key[j] = sw ? kb : ka; key[j + 1] = sw ? ka : kb;
        kk[j] = sw ? ib : ia; kk[j + 1] = sw ? ia : ib;
      }
    float ax = 3.0e38f, ay = 3.0e38f, az = 3.0e38f, bx = -3.0e38f, by = -3.0e38f, bz = -3.0e38f;
#pragma unroll
    for (int j = 0; j < SG; ++j) {
      const int k = kk[j];
      korig_w[lane * SG + j] = (unsigned short)k;
      px[j] = xyz[(size_t)k * 3 + 0];
      py[j] = xyz[(size_t)k * 3 + 1];
      pz[j] = xyz[(size_t)k * 3 + 2];
      asm volatile("" : "+v"(px[j]), "+v"(py[j]), "+v"(pz[j]));
      pt[j] = key[j] == 0xFFFFFFFFu ? -1.0f : 1e10f;       // a spare slot never exceeds a real point (min-distances >= 0)
      ax = d6_vmin(ax, px[j]); bx = d6_vmax(bx, px[j]);
      ay = d6_vmin(ay, py[j]); by = d6_vmax(by, py[j]);
      az = d6_vmin(az, pz[j]); bz = d6_vmax(bz, pz[j]);
    }
    lox = d6_wave_min(ax); hix = d6_wave_max(bx);
    loy = d6_wave_min(ay); hiy = d6_wave_max(by);
    loz = d6_wave_min(az); hiz = d6_wave_max(bz);
  }
  float cmax = __builtin_inff();

  int r_next = 0;                                         // next pick to apply = number of picks applied
  int idle = 0;
  while (r_next < m) {
    if (++idle > kIdleLimit) return;                      // watchdog (see the sequencer)
    // has the next pick been made?  (one broadcast read: 15 owners polling 64 entries each would take half the LDS bandwidth)
    if ((unsigned)(sq_ld(hist + (r_next & (kRing - 1)) * 3 + 2) >> 32) != (unsigned)r_next + 1u) {
      if (ow == 4) SQ_STAT(12, 1);
      // sleep until the sequencer's next s_wakeup (or ~2000 cycles, if the ping fell between the test and the sleep): a
      // polling owner must not take issue slots from the rescans and the sequencer on its SIMD
      for (int i = 0; i < tune_sleep; ++i) __builtin_amdgcn_s_sleep(8);
      continue;
    }
    // the next 64 picks at once
    const int e = r_next + lane;
    const u64 *hp = hist + (e & (kRing - 1)) * 3;
    const u64 wx = sq_ld(hp), wy = sq_ld(hp + 1), wz = sq_ld(hp + 2);
    const unsigned et = (unsigned)e + 1u;                 // ring tags are pick index + 1 (0 = never written)
    const bool valid = e < m && (unsigned)(wx >> 32) == et && (unsigned)(wy >> 32) == et && (unsigned)(wz >> 32) == et;
    const u64 vmask = __ballot(valid);
    const int nvalid = ~vmask == 0ull ? 64 : __builtin_ctzll(~vmask);
    if (nvalid == 0) continue;                            // (the z word was there, x or y not yet)
    if (ow == 4) SQ_STAT(14, 1);
    idle = 0;
    const float sx = sq_bitsf((unsigned)wx), sy = sq_bitsf((unsigned)wy), sz = sq_bitsf((unsigned)wz);
    const float gx = fmaxf(0.f, fmaxf(lox - sx, sx - hix));
    const float gy = fmaxf(0.f, fmaxf(loy - sy, sy - hiy));
    const float gz = fmaxf(0.f, fmaxf(loz - sz, sz - hiz));
    const float lb = d6_sqdist(gx, gy, gz);
    // picks of the batch that can change this wave's points (tested against the maximum BEFORE the batch: min-distances only
    // decrease, so the test stays conservative for the later ones)
    u64 need = __ballot(!(lb >= cmax)) & (nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull));
    r_next += nvalid;                                       // the whole batch is applied below: picks that fail the test change nothing
    if (need != 0ull) {
      // all of them in one go: plain min passes for all but the last, then ONE pass that also tracks the arg-max and
      // publishes the record — an owner that fell behind catches up at the price of one extraction
      while (need & (need - 1ull)) {
        const int i = __builtin_ctzll(need);
        need &= need - 1ull;
        sq_apply<SG>(d6_readlane_f(sx, i), d6_readlane_f(sy, i), d6_readlane_f(sz, i), px, py, pz, pt);
        if (ow == 4) SQ_STAT(15, 1);
      }
      const int i = __builtin_ctzll(need);
      if (ow == 4) SQ_STAT(13, 1);
      SQ_STAT(7, 1);
      cmax = sq_rescan<SG>(d6_readlane_f(sx, i), d6_readlane_f(sy, i), d6_readlane_f(sz, i), log2s, px, py, pz, pt, korig_w, rec, r_next);
    }
    if (lane == 0) __hip_atomic_store(&progress[ow], r_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// Lane 4j + s = candidate s of region j.
__device__ __forceinline__ void sq_sequencer(int m, int log2s, int idx_add, int *__restrict__ idxs, const u64 *rec, u64 *hist,
                                             const int *progress, unsigned long long *sq_stats_lds, int tune_wake) {
  const int lane = threadIdx.x & 63;
  const bool live = lane < kOwners * kCand;
  const int slot = lane & 3;
  const u64 *rp = rec + (live ? lane >> 2 : 0) * kRecStride;
  const u64 *rp_mine = rp + 5 * slot;
  int r_dec = 1;                                          // next pick to decide
  int minprog = 0;                                        // lower bound of every owner's progress
  unsigned a_tag = 0;                                     // tag of the accepted record (quad-uniform; 0 = none)
  float cv = -1.0f;                                       // current value of this candidate (-1: no candidate in this slot)
  float qx = 0.f, qy = 0.f, qz = 0.f;
  int kidx = 0;
  bool is_last = false;
  float bound_v = __builtin_inff();                       // value of the record's last candidate when it was made (quad-uniform)
  float hx = 0.f, hy = 0.f, hz = 0.f;                     // lane i & 63: pick i (the last 64 picks)
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
    // -- 1. new records?  (the owner writes the bound word last)
    const u64 wb = sq_ld(rp + kWordBound);
    const unsigned t0 = (unsigned)(wb >> 32);
    const bool changed = live && t0 > a_tag;
    if (__ballot(changed) != 0ull) {
      SQ_STAT(4, 1);
      const u64 wc = sq_ld(rp + kWordCount);
      const u64 wv = sq_ld(rp_mine), wk = sq_ld(rp_mine + 1), wx = sq_ld(rp_mine + 2), wy = sq_ld(rp_mine + 3), wz = sq_ld(rp_mine + 4);
      const int nc = (int)(unsigned)wc;
      const bool used = slot < nc;
      const bool mine_ok = (unsigned)(wc >> 32) == t0 &&
                           (!used || ((unsigned)(wv >> 32) == t0 && (unsigned)(wk >> 32) == t0 && (unsigned)(wx >> 32) == t0 &&
                                      (unsigned)(wy >> 32) == t0 && (unsigned)(wz >> 32) == t0));
      // a record is taken by all four lanes of its region or by none (a torn one — the owner is rewriting it — leaves the
      // region WITHOUT a record, unknown and unbounded, until the next poll: sound, and rare)
      float bad, dummy;
      sq_quad_max2(mine_ok ? 0.f : 1.f, 0.f, bad, dummy);
      const bool ok = changed && bad == 0.f;
      if (changed) {
        if (ok) {
          cv = used ? sq_bitsf((unsigned)wv) : -1.0f;
          kidx = (int)(unsigned)wk;
          qx = sq_bitsf((unsigned)wx); qy = sq_bitsf((unsigned)wy); qz = sq_bitsf((unsigned)wz);
          is_last = slot == nc - 1;
          bound_v = sq_bitsf((unsigned)wb);
          a_tag = t0;
        } else {
          bound_v = __builtin_inff();
          a_tag = 0u;
        }
      }
      if (__ballot(ok) != 0ull) {
        // picks made since those records: t0 .. r_dec-1
        const int from = (int)d6_wave_min(ok ? (float)t0 : 3.0e38f);
        SQ_STAT(5, r_dec - from);
        for (int i = from; i < r_dec; ++i) {
          float sx, sy, sz;
          if (r_dec - i <= 64) {
            sx = d6_readlane_f(hx, i & 63); sy = d6_readlane_f(hy, i & 63); sz = d6_readlane_f(hz, i & 63);
          } else {
            const u64 *hp = hist + (i & (kRing - 1)) * 3;
            sx = sq_bitsf((unsigned)sq_ld(hp)); sy = sq_bitsf((unsigned)sq_ld(hp + 1)); sz = sq_bitsf((unsigned)sq_ld(hp + 2));
          }
          if (ok && (int)t0 <= i) cv = d6_vmin(cv, d6_sqdist(qx - sx, qy - sy, qz - sz));
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
    // The region's maximum is exactly its best candidate X iff X is ordered before-or-at the record's last candidate as it
    // was: value above the bound, or equal to it with the last candidate itself untouched (then X is that candidate or one
    // ordered before it; an equal value reached by coincidence is treated as unknown — the rescan that lowered the last
    // candidate republishes the region).
    float xv, lv;
    sq_quad_max2(cv, is_last ? cv : -2.0f, xv, lv);
    const bool exact = live && a_tag != 0u && (xv > bound_v || (xv == bound_v && lv == bound_v));
    const float ev = exact ? cv : -1.0f;
    const float ub = live && !exact ? bound_v : -1.0f;
    float E, UB;
    sq_wave_max2(ev, ub, E, UB);
    if (!(UB < E)) { SQ_STAT(2, 1); SQ_STAT(10, SQ_CLK() - tq1); continue; }   // some region's maximum is unknown and may be the largest: poll
    const u64 tie = __ballot(ev == E);
    int wl = __builtin_ctzll(tie);
    if (__popcll(tie) != 1) wl = sq_min_key_lane(tie, sq_tie_key(kidx, log2s));
    const float sx = d6_readlane_f(qx, wl), sy = d6_readlane_f(qy, wl), sz = d6_readlane_f(qz, wl);
    const int k = d6_readlane_i(kidx, wl);
    if (lane < 3) sq_st(hist + (r_dec & (kRing - 1)) * 3 + lane, sq_pack(sq_fbits(lane == 0 ? sx : lane == 1 ? sy : sz), (unsigned)r_dec + 1u));
    if (lane == 0) idxs[r_dec] = k + idx_add;
    if (tune_wake) asm volatile("s_wakeup");               // owners asleep in their poll loop look at the ring now
    sq_writelane3(hx, hy, hz, sx, sy, sz, r_dec & 63);
    cv = d6_vmin(cv, d6_sqdist(qx - sx, qy - sy, qz - sz));   // (an empty slot stays at -1)
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

// One workgroup of 16 waves per scene.  `perm`: the scene's Morton permutation (n entries).
__global__ __launch_bounds__(1024) void fps_seq_kernel(int n, int m, int log2s, long long xyz_bstride, long long idx_bstride,
                                                       int idx_add, const float *__restrict__ xyz,
                                                       const int *__restrict__ perm, int *__restrict__ idxs, int tune_sleep,
                                                       int tune_wake) {
  __shared__ unsigned short korig[kOwners * kOwnerPoints];   // sorted slot -> original index
  __shared__ u64 rec[kOwners * kRecStride];
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
  for (int w = h; w < kOwners * kRecStride; w += 1024) rec[w] = 0ull;
  if (h < kOwners) progress[h] = 0;
  __syncthreads();                                        // the only barrier: the tags are clear
  // pick 0 is point 0 (sampling_gpu.cu:131-133): ring entry 0; every region's box test passes against its initial
  // maximum (+inf), so every owner starts with a rescan and publishes its first record (tag 1)
  if (h < 3) sq_st(hist + h, sq_pack(sq_fbits(xyz[h]), 1u));
  if (h == 0) idxs[0] = idx_add;
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    sq_sequencer(m, log2s, idx_add, idxs, rec, hist, progress, sq_stats, tune_wake);
    return;
  }
  sq_owner(wave - 1, n, m, log2s, xyz, perm, korig, rec, hist, progress, sq_stats, tune_sleep);
}

}  // namespace

#ifdef DET6D_EXPERIMENTS
// 0 sequencer steps, 1 decisions, 2 blocked polls, 3 ring waits, 4 polls that saw a new record, 5 picks replayed onto new
// records, 7 rescans (all owners), 8 cycles polling / accepting, 9 cycles deciding, 11 total cycles,
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
  (void)regions_per_wave;
  static const int tune_sleep = det6d_env_int("DET6D_FPS_SEQ_SLEEP", 4);   // x 512 cycles
  static const int tune_wake = det6d_env_int("DET6D_FPS_SEQ_WAKE", 1);
  hipLaunchKernelGGL(fps_seq_kernel, dim3(b), dim3(1024), 0, stream, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz, perm, idx,
                     tune_sleep, tune_wake);
  return det6d_check_launch("det6d_fps (look-ahead)");
}
