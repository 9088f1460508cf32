// fps_coop.hip — farthest point sampling (D-FPS) of clouds too large for one CU's register file
// (N = 32768 / 65536: BASELINE.json configs[4], the 65536-point scenes), bit for bit the picks of
// farthest_point_sampling_kernel (core/pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:101-222, tie order
// of its shared-memory tree :94-99,159-216).
//
// The memory-resident fallback (fps.hip: fps_mem_kernel) re-reads all N points from L2 every round: 111 us per round
// on MI355X, 1.8 s for the 16383 rounds of a 65536-point scene.  Here a scene is held IN REGISTERS by PARTS = N / 16384
// cooperating workgroups (one per CU): the scene is sorted along a Morton curve (device radix sort over all scenes of
// the launch, key = scene | 20-bit Morton code), part q keeps sorted positions [q * 16384, (q+1) * 16384) exactly like
// the wave-skip sampler of fps_cells.hip (16 waves x 16 points per lane, one bounding box per wave, cached arg-max,
// exact floating-point skip test, explicit tie paths).  Per round every part reduces to ITS best candidate as before
// (one workgroup barrier), publishes it in a 40-byte global slot, and every wave of every part polls the PARTS slots
// of the round and takes the global arg-max under the reference's tie key — no second barrier, no host involvement.
//
// Exchange protocol (no fences: every 8-byte word carries its own round tag):
//   slot(scene, parity, part) = 5 words {payload, tag}: value, original index, x, y, z.  Words are written / read with
//   relaxed agent-scope 64-bit atomics (sc1: visible across XCDs); a reader accepts a slot when all five tags equal
//   the round.  Two parities suffice: a part that writes round r+1 has read every slot of round r, so every part has
//   written round r and is done reading round r-1.  The area is zeroed by the key kernel of the same launch (stream
//   order), tags are round numbers >= 1.
// Placement: block ids of one scene are congruent mod 8, i.e. on one XCD under round-robin dispatch (speed only).
// Publishing stores are AGENT scope (the HIP memory model's guarantee that another workgroup's agent-scope load sees them).
// DET6D_FPS_COOP_FAST=1 allows a part to publish with workgroup-scope stores — they stay in the XCD's L2 instead of being
// written through to the fabric: 1.63 -> 1.40 us per round — when (a) all parts of its scene report the same HW_REG_XCC_ID
// and (b) a handshake in round 0 has shown, on this device and this placement, that a workgroup-scope store of every part
// reaches the other parts' agent-scope loads; a part that fails either test keeps agent scope (mixing is fine: readers
// always load with agent scope).
// A part that waits longer than ~2 s for a partner (it can only be a scheduling accident) raises the error word of
// the workspace and leaves: det6d_fps_fused_status reports the launch as failed instead of the GPU hanging.
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace {

constexpr int kPartPoints = 16384;
constexpr int kSlotWords = 8;   // 5 used, padded to 64 bytes

__device__ __forceinline__ unsigned co_bitrev_bits(unsigned v, int bits) {
  return bits == 0 ? 0u : (__builtin_bitreverse32(v) >> (32 - bits));
}
// order key of point k under the reference's tie rule (smaller wins): (bitrev_{log2 S}(k mod S), k)
__device__ __forceinline__ unsigned co_tie_key(int k, int log2s) {
  return (co_bitrev_bits((unsigned)k & ((1u << log2s) - 1u), log2s) << (32 - log2s)) | ((unsigned)k >> log2s);
}
__device__ __forceinline__ int co_min_key_lane(unsigned long long cand, int k, int log2s) {
  const int lane = threadIdx.x & 63;
  const bool mine = (cand >> lane) & 1ull;
  unsigned key = mine ? co_tie_key(k, log2s) : 0xFFFFFFFFu;
  unsigned m = key;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)m, off);
    m = o < m ? o : m;
  }
  return __builtin_ctzll(__ballot(mine && key == m));
}
__device__ __forceinline__ unsigned co_part1by1(unsigned v) {
  v &= 0xFFFFu;
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

// ---- pre-pass 1: per-scene (x, y) extent -> 20-bit Morton keys, scene index in the bits above; clears the exchange area
__global__ __launch_bounds__(1024) void coop_keys_kernel(int n, long long xyz_bstride, const float *__restrict__ xyz,
                                                         unsigned *__restrict__ keys, unsigned *__restrict__ vals,
                                                         unsigned long long *__restrict__ exch, int exch_words, int *err) {
  __shared__ float red[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, scene = blockIdx.x;
  const float *p = xyz + (size_t)scene * xyz_bstride;
  float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
  for (int k = tid; k < n; k += 1024) {
    const float x = p[(size_t)k * 3], y = p[(size_t)k * 3 + 1];
    if (x == x && fabsf(x) < 1e30f) { xmin = fminf(xmin, x); xmax = fmaxf(xmax, x); }
    if (y == y && fabsf(y) < 1e30f) { ymin = fminf(ymin, y); ymax = fmaxf(ymax, y); }
  }
  xmin = d6_wave_min(xmin); xmax = d6_wave_max(xmax); ymin = d6_wave_min(ymin); ymax = d6_wave_max(ymax);
  if (lane == 0) { red[0][wave] = xmin; red[1][wave] = xmax; red[2][wave] = ymin; red[3][wave] = ymax; }
  __syncthreads();
  xmin = red[0][0]; xmax = red[1][0]; ymin = red[2][0]; ymax = red[3][0];
  for (int w = 1; w < 16; ++w) {
    xmin = fminf(xmin, red[0][w]); xmax = fmaxf(xmax, red[1][w]);
    ymin = fminf(ymin, red[2][w]); ymax = fmaxf(ymax, red[3][w]);
  }
  const float sx = xmax > xmin ? 1023.0f / (xmax - xmin) : 0.f;
  const float sy = ymax > ymin ? 1023.0f / (ymax - ymin) : 0.f;
  for (int k = tid; k < n; k += 1024) {
    float fx = (p[(size_t)k * 3] - xmin) * sx, fy = (p[(size_t)k * 3 + 1] - ymin) * sy;
    fx = fx == fx ? fminf(fmaxf(fx, 0.f), 1023.f) : 0.f;
    fy = fy == fy ? fminf(fmaxf(fy, 0.f), 1023.f) : 0.f;
    keys[(size_t)scene * n + k] = ((unsigned)scene << 20) | (co_part1by1((unsigned)fx) << 1) | co_part1by1((unsigned)fy);
    vals[(size_t)scene * n + k] = (unsigned)k;
  }
  for (int w = tid; w < exch_words; w += 1024) exch[(size_t)scene * exch_words + w] = 0ull;
  // the error word is STICKY: it is cleared when the workspace is created (zero-filled by the caller) and by
  // det6d_fps_fused_status once it has been read, never by a launch — a later launch must not hide an earlier failure
  (void)err;
}

// ---- pre-pass 2: the 16 points of a lane ordered by the reference's tie key (strict '>' of the scan keeps the right one)
__global__ __launch_bounds__(512) void coop_group_order_kernel(long long total_groups, int n, int log2s, unsigned *__restrict__ perm) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total_groups) return;
  unsigned *p = perm + g * 16;
  unsigned v[16], key[16];
  for (int i = 0; i < 16; ++i) { v[i] = p[i]; key[i] = co_tie_key((int)v[i], log2s); }
  for (int i = 1; i < 16; ++i) {
    const unsigned vi = v[i], ki = key[i];
    int j = i;
    while (j > 0 && key[j - 1] > ki) { v[j] = v[j - 1]; key[j] = key[j - 1]; --j; }
    v[j] = vi; key[j] = ki;
  }
  for (int i = 0; i < 16; ++i) p[i] = v[i];
}

template <int LO, int HI, int N>
__device__ __forceinline__ void co_pick(int ws, int wl, const float (&px)[N], const float (&py)[N], const float (&pz)[N],
                                        float &sx, float &sy, float &sz) {
  if constexpr (HI - LO == 1) {
    sx = d6_readlane_f(px[LO], wl);
    sy = d6_readlane_f(py[LO], wl);
    sz = d6_readlane_f(pz[LO], wl);
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (ws < MID) co_pick<LO, MID>(ws, wl, px, py, pz, sx, sy, sz);
    else co_pick<MID, HI>(ws, wl, px, py, pz, sx, sy, sz);
  }
}

typedef float co_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long co_pack(unsigned payload, unsigned tag) {
  return ((unsigned long long)tag << 32) | payload;
}

template <int PARTS>
__global__ __launch_bounds__(1024) void fps_coop_kernel(int b, int n, int m, int log2s, long long xyz_bstride,
                                                        long long idx_bstride, int idx_add, const float *__restrict__ xyz,
                                                        const unsigned *__restrict__ perm, int *__restrict__ idxs,
                                                        unsigned long long *__restrict__ exch, int *err, int allow_fast) {
  constexpr int NW = 16, SLOTS = 16, HG = SLOTS / 2;
  __shared__ float4 slot_v[2][NW];
  __shared__ int slot_k[2][NW];
  __shared__ unsigned short korig[64 * NW * SLOTS];
  const int h = threadIdx.x, lane = h & 63, wave = h >> 6;
  // block -> (scene, part): the parts of a scene sit on block ids congruent mod 8 (one XCD under round-robin dispatch)
  const int t = blockIdx.x >> 3;
  const int part = t % PARTS;
  const int scene = (t / PARTS) * 8 + (blockIdx.x & 7);
  if (scene >= b) return;
  xyz += (size_t)scene * xyz_bstride;
  perm += (size_t)scene * n + (size_t)part * kPartPoints;
  idxs += (size_t)scene * idx_bstride;
  exch += (size_t)scene * (2 * PARTS * kSlotWords);

  float px[SLOTS], py[SLOTS], pz[SLOTS], pt[SLOTS];
  float lox, loy, loz, hix, hiy, hiz;
  {
    float ax = 3.0e38f, ay = 3.0e38f, az = 3.0e38f, bx = -3.0e38f, by = -3.0e38f, bz = -3.0e38f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int pos = (wave * 64 + lane) * SLOTS + s;
      const int k = (int)perm[pos];
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
    lox = d6_wave_min(ax); hix = d6_wave_max(bx);
    loy = d6_wave_min(ay); hiy = d6_wave_max(by);
    loz = d6_wave_min(az); hiz = d6_wave_max(bz);
  }
  __syncthreads();

  float cx = xyz[0], cy = xyz[1], cz = xyz[2];      // the first pick is point 0 (sampling_gpu.cu:131-133)
  if (part == 0 && h == 0) idxs[0] = idx_add;
  // Round 0 of the exchange: every part publishes the XCD it runs on (agent-scope words, valid wherever the parts sit).
  // With DET6D_FPS_COOP_FAST=1, parts that share one XCD — the placement the block numbering aims for — AND pass the
  // handshake below publish later rounds with workgroup-scope stores that STAY in that XCD's L2, where the partners'
  // L1-bypassing loads find them at L2-hit latency; agent-scope stores (the default) are written through to the fabric, so
  // every poll pays a memory-side round trip.
  bool same_xcd;
  {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;          // hwreg(HW_REG_XCC_ID, 0, 4)
    unsigned long long *mine0 = exch + (size_t)part * kSlotWords + 5;               // word 5 of the parity-0 slot
    if (h == 0) __hip_atomic_store(mine0, co_pack(xcc, 0x7fffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool reader0 = lane < PARTS;
    const unsigned long long *theirs0 = exch + (size_t)(reader0 ? lane : 0) * kSlotWords + 5;
    unsigned long long w0 = 0ull;
    int spins0 = 0;
    for (;;) {
      w0 = __hip_atomic_load(theirs0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool ok = !reader0 || (unsigned)(w0 >> 32) == 0x7fffffffu;
      if (__ballot(!ok) == 0ull) break;
      if (++spins0 > (1 << 22)) {
        if (lane == 0) atomicExch(err, 1);
        if (part == 0) for (int i = 1 + h; i < m; i += 1024) idxs[i] = idx_add;   // in-range picks: nothing downstream may fault
        return;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    same_xcd = allow_fast && __ballot(reader0 && (unsigned)w0 != xcc) == 0ull;
    if (allow_fast) {
      // handshake: does a WORKGROUP-scope store of every part reach this part's agent-scope loads?  (word 6 of the parity-0
      // slots; bounded wait: a part that does not see all tokens publishes with agent scope)
      unsigned long long *tok = exch + (size_t)part * kSlotWords + 6;
      if (h == 0) __hip_atomic_store(tok, co_pack(0x5A5A0000u | (unsigned)part, 0x7ffffffeu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned long long *their_tok = exch + (size_t)(reader0 ? lane : 0) * kSlotWords + 6;
      bool seen_all = false;
      for (int spins1 = 0; spins1 < 4096 && !seen_all; ++spins1) {
        const unsigned long long w1 = __hip_atomic_load(their_tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen_all = __ballot(reader0 && (unsigned)(w1 >> 32) != 0x7ffffffeu) == 0ull;
        if (!seen_all) __builtin_amdgcn_s_sleep(1);
      }
      same_xcd = same_xcd && seen_all;
    }
  }
  float cg_val = __builtin_inff(), cg_x = 0.f, cg_y = 0.f, cg_z = 0.f;
  int cg_k = 0;

  for (int r = 1; r < m; ++r) {
    // 1. can any point of this wave change?  (exact floating-point lower bound of the scan's distance)
    const float gx = fmaxf(0.f, fmaxf(lox - cx, cx - hix));
    const float gy = fmaxf(0.f, fmaxf(loy - cy, cy - hiy));
    const float gz = fmaxf(0.f, fmaxf(loz - cz, cz - hiz));
    const float lb = d6_sqdist(gx, gy, gz);
    if (!(lb >= cg_val)) {
      float best = -1.0f;
      int bs = 0;
      const co_f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
      for (int q = 0; q < HG; ++q) {
        const int s0 = 2 * q;
        const co_f32x2 dx = co_f32x2{px[s0], px[s0 + 1]} - c2x;
        const co_f32x2 dy = co_f32x2{py[s0], py[s0 + 1]} - c2y;
        const co_f32x2 dz = co_f32x2{pz[s0], pz[s0 + 1]} - c2z;
        co_f32x2 d = dy * dy;
        d = __builtin_elementwise_fma(dx, dx, d);
        d = __builtin_elementwise_fma(dz, dz, d);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int s = s0 + e;
          const float tt = d6_vmin(d[e], pt[s]);
          pt[s] = tt;
          const bool up = tt > best;
          bs = up ? s : bs;
          best = up ? tt : best;
        }
      }
      const float wmax = d6_wave_max(best);
      const unsigned long long tie = __ballot(best == wmax);
      int wl = __builtin_ctzll(tie);
      const int gbase = wave * 64;
      if (__popcll(tie) != 1) wl = co_min_key_lane(tie, (int)korig[(gbase + lane) * SLOTS + bs], log2s);
      const int ws = d6_readlane_i(bs, wl);
      cg_val = wmax;
      cg_k = (int)korig[(gbase + wl) * SLOTS + ws];
      co_pick<0, SLOTS>(ws, wl, px, py, pz, cg_x, cg_y, cg_z);
    }
    // 2. this part's arg-max over its waves' cached maxima
    if (lane == 0) {
      slot_v[r & 1][wave] = make_float4(cg_val, cg_x, cg_y, cg_z);
      slot_k[r & 1][wave] = cg_k;
    }
    __syncthreads();
    const int src = lane & (NW - 1);
    const float4 e2 = slot_v[r & 1][src];
    const int i2 = slot_k[r & 1][src];
    const float bmax = d6_row_max16(e2.x);
    const unsigned long long tie2 = __ballot(e2.x == bmax) & ((1ull << NW) - 1ull);
    int ww = __builtin_ctzll(tie2);
    if (__popcll(tie2) != 1) ww = co_min_key_lane(tie2, i2, log2s);
    const int pk = d6_readlane_i(i2, ww);
    const float pxw = d6_readlane_f(e2.y, ww), pyw = d6_readlane_f(e2.z, ww), pzw = d6_readlane_f(e2.w, ww);
    // 3. publish (wave 0), then every wave polls the PARTS slots of this round
    unsigned long long *mine = exch + ((size_t)(r & 1) * PARTS + part) * kSlotWords;
    if (wave == 0 && lane < 5) {
      const unsigned payload = lane == 0 ? __builtin_bit_cast(unsigned, bmax) : lane == 1 ? (unsigned)pk
                               : lane == 2 ? __builtin_bit_cast(unsigned, pxw) : lane == 3 ? __builtin_bit_cast(unsigned, pyw)
                                                                                           : __builtin_bit_cast(unsigned, pzw);
      if (same_xcd) __hip_atomic_store(mine + lane, co_pack(payload, (unsigned)r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_store(mine + lane, co_pack(payload, (unsigned)r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // lane 5 * p + w reads word w of part p
    const int rp = lane / 5, rw = lane - 5 * rp;
    const bool reader = lane < 5 * PARTS;
    const unsigned long long *theirs = exch + ((size_t)(r & 1) * PARTS + (reader ? rp : 0)) * kSlotWords + (reader ? rw : 0);
    unsigned long long word = 0ull;
    int spins = 0;
    bool dead = false;
    for (;;) {
      word = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool ok = !reader || (unsigned)(word >> 32) == (unsigned)r;
      if (__ballot(!ok) == 0ull) break;
      if (++spins > (1 << 22)) { dead = true; break; }   // ~2 s: a partner never arrived
      __builtin_amdgcn_s_sleep(1);
    }
    if (dead) {
      if (lane == 0) atomicExch(err, 1);
      if (part == 0) for (int i = r + h; i < m; i += 1024) idxs[i] = idx_add;     // in-range picks: nothing downstream may fault
      return;
    }
    const unsigned pay = (unsigned)word;
    // gather the slots: lane p (< PARTS) gets part p's candidate
    const int pl = lane < PARTS ? lane : 0;
    const float v2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (5 * pl + 0), (int)pay));
    const int k2 = __builtin_amdgcn_ds_bpermute(4 * (5 * pl + 1), (int)pay);
    const float x2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (5 * pl + 2), (int)pay));
    const float y2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (5 * pl + 3), (int)pay));
    const float z2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (5 * pl + 4), (int)pay));
    const float vv = lane < PARTS ? v2 : -__builtin_inff();
    const float gmax = d6_row_max16(vv);
    const unsigned long long tie3 = __ballot(vv == gmax) & ((1ull << PARTS) - 1ull);
    int wp = __builtin_ctzll(tie3);
    if (__popcll(tie3) != 1) wp = co_min_key_lane(tie3, k2, log2s);
    const int old = d6_readlane_i(k2, wp);
    cx = d6_readlane_f(x2, wp);
    cy = d6_readlane_f(y2, wp);
    cz = d6_readlane_f(z2, wp);
    if (part == 0 && h == 0) idxs[r] = old + idx_add;
  }
}

struct CoopLayout {
  size_t keys_in, keys_out, vals_in, vals_out, cub, exch, err, total;
  size_t cub_bytes;
};

int scene_bits(int b) {
  int bits = 0;
  while ((1 << bits) < b) ++bits;
  return bits;
}

CoopLayout coop_layout(int b, int n) {
  CoopLayout L;
  const size_t items = (size_t)b * n;
  auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t cub = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, cub, (const unsigned *)nullptr, (unsigned *)nullptr, (const unsigned *)nullptr,
                                     (unsigned *)nullptr, (unsigned)items, 0, 20 + scene_bits(b), (hipStream_t)0);
  L.cub_bytes = cub;
  size_t off = 0;
  L.err = off; off = align(off + 256);          // first: its offset must not depend on b (a workspace serves launches of fewer scenes)
  L.keys_in = off; off = align(off + items * 4);
  L.keys_out = off; off = align(off + items * 4);
  L.vals_in = off; off = align(off + items * 4);
  L.vals_out = off; off = align(off + items * 4);
  L.cub = off; off = align(off + cub);
  L.exch = off; off = align(off + (size_t)b * 2 * 4 * kSlotWords * 8);
  L.total = off;
  return L;
}

}  // namespace

// does the cooperative sampler take (n, fresh min-distances)?
bool det6d_fps_coop_handles(int n) { return n == 2 * kPartPoints || n == 4 * kPartPoints; }

long long det6d_fps_coop_workspace_bytes(int b, int n) {
  if (!det6d_fps_coop_handles(n) || b <= 0 || b > 4096) return 0;
  return (long long)coop_layout(b, n).total;
}

// D-FPS of b scenes of n = 32768 / 65536 points with fresh min-distances; `workspace` of
// det6d_fps_coop_workspace_bytes(b, n) bytes (256-byte aligned)
int det6d_fps_coop_launch(int b, int n, int m, int log2s, long long xyz_bstride, long long idx_bstride, int idx_add,
                          const float *xyz, void *workspace, int *idx, hipStream_t stream) {
  if (!det6d_fps_coop_handles(n) || b <= 0 || b > 4096 || !workspace || ((uintptr_t)workspace & 255)) return DET6D_EINVAL;
  const CoopLayout L = coop_layout(b, n);
  char *ws = (char *)workspace;
  unsigned *keys_in = (unsigned *)(ws + L.keys_in), *keys_out = (unsigned *)(ws + L.keys_out);
  unsigned *vals_in = (unsigned *)(ws + L.vals_in), *vals_out = (unsigned *)(ws + L.vals_out);
  unsigned long long *exch = (unsigned long long *)(ws + L.exch);
  int *err = (int *)(ws + L.err);
  const int parts = n / kPartPoints;
  hipLaunchKernelGGL(coop_keys_kernel, dim3(b), dim3(1024), 0, stream, n, xyz_bstride, xyz, keys_in, vals_in, exch,
                     2 * parts * kSlotWords, err);
  size_t cub = L.cub_bytes;
  const hipError_t sort_rc = hipcub::DeviceRadixSort::SortPairs(ws + L.cub, cub, keys_in, keys_out, vals_in, vals_out,
                                                                (unsigned)((size_t)b * n), 0, 20 + scene_bits(b), stream);
  if (sort_rc != hipSuccess) {
    det6d_set_error("det6d_fps (cooperative: radix sort)", sort_rc);
    return DET6D_ELAUNCH;
  }
  const long long groups = (long long)b * n / 16;
  hipLaunchKernelGGL(coop_group_order_kernel, dim3((unsigned)((groups + 511) / 512)), dim3(512), 0, stream, groups, n, log2s, vals_out);
  const int grid = 8 * parts * ((b + 7) / 8);
  // DET6D_FPS_COOP_FAST=1: workgroup-scope publishing stores where the placement test and the handshake allow (see the top)
  static const int allow_fast = det6d_switch_int("DET6D_FPS_COOP_FAST", 0) ? 1 : 0;
  if (parts == 4)
    hipLaunchKernelGGL(fps_coop_kernel<4>, dim3(grid), dim3(1024), 0, stream, b, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz,
                       vals_out, idx, exch, err, allow_fast);
  else
    hipLaunchKernelGGL(fps_coop_kernel<2>, dim3(grid), dim3(1024), 0, stream, b, n, m, log2s, xyz_bstride, idx_bstride, idx_add, xyz,
                       vals_out, idx, exch, err, allow_fast);
  return det6d_check_launch("det6d_fps (cooperative)");
}

// byte offset of the (sticky) error word inside a cooperative workspace
long long det6d_fps_coop_status_offset(int b, int n) {
  if (!det6d_fps_coop_handles(n) || b <= 0 || b > 4096) return -1;
  return (long long)coop_layout(b, n).err;
}

// error word of the cooperative launches on `workspace` since it was last read (synchronises `stream`): 0 = fine.  Reading a
// set word clears it.
int det6d_fps_coop_status(int b, int n, const void *workspace, hipStream_t stream) {
  if (!det6d_fps_coop_handles(n) || b <= 0 || b > 4096 || !workspace) return DET6D_EINVAL;
  const CoopLayout L = coop_layout(b, n);
  int flag = 0;
  if (hipMemcpyAsync(&flag, (const char *)workspace + L.err, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)
    return DET6D_ELAUNCH;
  if (flag) {
    hipMemsetAsync((char *)workspace + L.err, 0, sizeof(int), stream);
    det6d_set_error("det6d_fps (cooperative): a workgroup waited > 2 s for its partners", hipErrorLaunchFailure);
    return DET6D_ELAUNCH;
  }
  return DET6D_OK;
}
