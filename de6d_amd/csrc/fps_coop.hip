// fps_coop.hip — farthest point sampling (D-FPS) of clouds too large for one CU's register file
// (N = 32768 / 65536: BASELINE.json configs[4], the 65536-point scenes), bit for bit the picks of
// farthest_point_sampling_kernel (core/pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:101-222, tie order of its
// shared-memory tree :94-99,159-216).
//
// The memory-resident fallback (fps.hip: fps_mem_kernel) re-reads all N points from L2 every pick: 111 us per pick on MI355X,
// 1.8 s for the 16383 picks of a 65536-point scene.  Here a scene is held IN REGISTERS by PARTS = N / 16384 cooperating
// workgroups (one per CU): the scene is cut into PARTS spatial parts of 16384 points (coop_split_kernel: counting partitions by
// x and y), every part is put into the 4 x 4 k-d order of the single-workgroup sampler (fps_cells.hip: cell_sort_kernel), and
// part q keeps its 16384 points as 16 waves x 16 points per lane, one bounding box per wave, exact floating-point skip test,
// explicit tie paths — the wave-skip sampler of fps_cells.hip.
//
// Shipped form: MULTI-PICK rounds (fps_coop_multi_kernel; decision rule and its executable model: fps_seq.hip,
// tests/models/fps_lookahead.py).  Per round every wave applies the round's picks to its points and republishes its top 4
// points in LDS when they could change; wave 0 of every part publishes the part's 64 candidates as tagged 8-byte words, gathers
// the candidates of all parts and decides as many picks as the rule allows (24-28 per round on the benchmark scenes, cap 64);
// every part takes the same decisions from the same words, so one L2 round trip serves a whole round.  0.85 us per pick
// against 1.63 for the one-pick form (rounds 2-3, experiments build: DET6D_FPS_COOP_MULTI=0), which exchanged one candidate
// per part and per pick.
//
// Exchange protocol (no fences: every 8-byte word carries its own round tag):
//   words are written / read with relaxed 64-bit atomics; a reader accepts a candidate when its five tags equal the round.
//   Two parities suffice: a part that writes round r+1 has read every word of round r, so every part has written round r and
//   is done reading round r-1.  The area is zeroed by the key kernel of the same launch (stream order), tags are >= 1.
// Publishing stores are AGENT scope (the HIP memory model's guarantee that another workgroup's agent-scope load sees them).
// DET6D_FPS_COOP_FAST=1 allows a part to publish with workgroup-scope stores — they stay in the XCD's L2 instead of being
// written through to the fabric — when (a) all parts of its scene report the same HW_REG_XCC_ID and (b) a handshake in round
// 0 has shown, on this device and this placement, that a workgroup-scope store of every part reaches this part's
// agent-scope loads; a part that fails either test keeps agent scope (mixing is fine: readers always load with agent scope).
// Placement: block ids of one scene are congruent mod 8, i.e. on one XCD under round-robin dispatch (speed only).
// A part that waits longer than ~2 s for a partner (it can only be a scheduling accident) raises the error word of the
// workspace, completes the scene's picks with a valid index and leaves: det6d_fps_fused_status reports the launch as failed
// instead of the GPU hanging.
#include "fps_multi.h"

namespace {

constexpr int kPartPoints = 16384;
constexpr int kSlotWords = 8;   // 5 used, padded to 64 bytes
constexpr int kMultiCands = 4;  // candidates per wave record of the multi-pick kernel

// 64-bit words of one scene's exchange area in the multi-pick kernel: round-0 words + two parities of candidate words
__host__ __device__ constexpr size_t coop_multi_words(int parts, int k) { return (size_t)8 * parts + (size_t)2 * parts * 16 * k * 5; }

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
// ---- pre-pass 1: the scene cut into PARTS spatial parts of exactly 16384 points (PARTS = 4: two halves by x, each cut in two
// by y; PARTS = 2: two halves by x); clears the exchange area.  Counting partition in LDS: a histogram over 4096 bins of the
// coordinate, an exclusive scan, one returning atomic per point for its position; the position decides the half — exact equal
// counts whatever the distribution, the order inside a bin is whatever the atomics make it (any partition into equal parts
// is CORRECT for the sampler; compact parts make its bounding boxes tight).  part_idx[scene][part][0 .. 16383] = the part's
// points, which cell_sort_kernel<16> (fps_cells.hip) then puts into the 4 x 4 k-d order of the single-workgroup sampler.
// (Rounds 2-4: 20-bit Morton keys + hipcub::DeviceRadixSort over all scenes + a lane-ordering kernel: 8 launches.)
constexpr int kSplitBins = 4096;

__device__ __forceinline__ unsigned co_block_exclusive_sum(unsigned v, unsigned *__restrict__ wtot) {
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

template <int PARTS>
__global__ __launch_bounds__(1024) void coop_split_kernel(int n, long long xyz_bstride, const float *__restrict__ xyz,
                                                          int *__restrict__ part_idx, unsigned long long *__restrict__ exch,
                                                          int exch_words) {
  constexpr int IPT = PARTS * kPartPoints / 1024;      // points per thread
  __shared__ float red[4][16];
  __shared__ unsigned wtot[16];
  __shared__ unsigned bins[2 * kSplitBins];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, scene = blockIdx.x;
  const float *p = xyz + (size_t)scene * xyz_bstride;
  part_idx += (size_t)scene * n;
  float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
  for (int k = tid; k < n; k += 1024) {
    const float x = p[(size_t)k * 3], y = p[(size_t)k * 3 + 1];
    if (x == x && fabsf(x) < 1e30f) { xmin = fminf(xmin, x); xmax = fmaxf(xmax, x); }
    if (y == y && fabsf(y) < 1e30f) { ymin = fminf(ymin, y); ymax = fmaxf(ymax, y); }
  }
  xmin = d6_wave_min(xmin); xmax = d6_wave_max(xmax); ymin = d6_wave_min(ymin); ymax = d6_wave_max(ymax);
  if (lane == 0) { red[0][wave] = xmin; red[1][wave] = xmax; red[2][wave] = ymin; red[3][wave] = ymax; }
  for (int c = tid; c < 2 * kSplitBins; c += 1024) bins[c] = 0u;
  __syncthreads();
  xmin = red[0][0]; xmax = red[1][0]; ymin = red[2][0]; ymax = red[3][0];
  for (int w = 1; w < 16; ++w) {
    xmin = fminf(xmin, red[0][w]); xmax = fmaxf(xmax, red[1][w]);
    ymin = fminf(ymin, red[2][w]); ymax = fmaxf(ymax, red[3][w]);
  }
  const float sx = xmax > xmin ? (float)(kSplitBins - 1) / (xmax - xmin) : 0.f;
  const float sy = ymax > ymin ? (float)(kSplitBins - 1) / (ymax - ymin) : 0.f;
  auto bin_of = [](float v, float lo, float scale) {
    float f = (v - lo) * scale;
    f = f == f ? fminf(fmaxf(f, 0.f), (float)(kSplitBins - 1)) : 0.f;
    return (unsigned)f;
  };
  // ---- x: position of every point in x order -> half
  for (int i = 0; i < IPT; ++i) atomicAdd(&bins[bin_of(p[(size_t)(i * 1024 + tid) * 3], xmin, sx)], 1u);
  __syncthreads();
  {
    unsigned c[4], total = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) { c[q] = bins[tid * 4 + q]; total += c[q]; }
    unsigned base = co_block_exclusive_sum(total, wtot);
#pragma unroll
    for (int q = 0; q < 4; ++q) { bins[tid * 4 + q] = base; base += c[q]; }
  }
  __syncthreads();
  unsigned long long upper = 0ull;                     // bit i: point i of this thread lies in the upper half by x
  for (int i = 0; i < IPT; ++i) {
    const int k = i * 1024 + tid;
    const unsigned pos = atomicAdd(&bins[bin_of(p[(size_t)k * 3], xmin, sx)], 1u);    // 0 .. n - 1, each once
    if (PARTS == 2) part_idx[pos] = k;                 // part = pos / 16384, slot = pos % 16384
    else if (pos >= (unsigned)(n / 2)) upper |= 1ull << i;
  }
  if (PARTS == 4) {
    // ---- y inside the halves: bins [half][4096]; a half holds n / 2 points exactly, so its positions start at half * n / 2
    __syncthreads();
    for (int c = tid; c < 2 * kSplitBins; c += 1024) bins[c] = 0u;
    __syncthreads();
    for (int i = 0; i < IPT; ++i) {
      const unsigned half = (unsigned)((upper >> i) & 1ull);
      atomicAdd(&bins[half * kSplitBins + bin_of(p[(size_t)(i * 1024 + tid) * 3 + 1], ymin, sy)], 1u);
    }
    __syncthreads();
    {
      unsigned c[8], total = 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q) { c[q] = bins[tid * 8 + q]; total += c[q]; }
      unsigned base = co_block_exclusive_sum(total, wtot);
#pragma unroll
      for (int q = 0; q < 8; ++q) { bins[tid * 8 + q] = base; base += c[q]; }
    }
    __syncthreads();
    for (int i = 0; i < IPT; ++i) {
      const int k = i * 1024 + tid;
      const unsigned half = (unsigned)((upper >> i) & 1ull);
      const unsigned pos = atomicAdd(&bins[half * kSplitBins + bin_of(p[(size_t)k * 3 + 1], ymin, sy)], 1u);
      part_idx[pos] = k;                               // positions [q 16384, (q + 1) 16384) = part q (x half, then y half)
    }
  }
  for (int w = tid; w < exch_words; w += 1024) exch[(size_t)scene * exch_words + w] = 0ull;
  // (the error word of the workspace is STICKY: it is cleared when the workspace is created (zero-filled by the caller) and by
  // det6d_fps_fused_status once it has been read, never by a launch — a later launch must not hide an earlier failure)
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

#ifdef DET6D_EXPERIMENTS   // the one-pick form (rounds 2-3): kept for A/B runs, DET6D_FPS_COOP_MULTI=0
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

#endif  // DET6D_EXPERIMENTS

// ------------------------------------------------------------------------------------------------------------------------
// Multi-pick form.  The kernel above pays one L2 exchange (0.55 us agent scope) and one rescan on the critical path of EVERY
// pick.  Here a round is: (A) every wave applies the picks of the last round to its 1024 points and, if they could change
// anything, rewrites its record — its top K points in the reference's order — in LDS; (B) wave 0 of every part publishes the
// part's 16 K candidates as tagged 8-byte words, polls the candidates of ALL parts of the scene, and decides as many picks as
// the records allow with the rule of fps_seq.hip (the best exact candidate, provided every unknown region's bound is strictly
// below it).  Every part runs the same decisions on the same records, so the picks need no second exchange: one L2 round trip
// per ROUND, typically 8-12 picks.  Sequencer lane l holds candidate l of set s = 0 .. SETS-1 (global candidate 64 s + l;
// a wave's K candidates are K neighbouring lanes of one set).
template <int PARTS, int K>
__global__ __launch_bounds__(1024) void fps_coop_multi_kernel(int b, int n, int m, int log2s, long long xyz_bstride,
                                                              long long idx_bstride, int idx_add, const float *__restrict__ xyz,
                                                              const unsigned *__restrict__ perm, int *__restrict__ idxs,
                                                              unsigned long long *__restrict__ exch, int *err, int allow_fast,
                                                              int max_picks) {
  constexpr int SG = kSlots;
  constexpr int CPP = kWaves * K;                       // candidates per part
  constexpr int SETS = PARTS * CPP / 64;                // candidates per sequencer lane
  constexpr int LOG2K = K == 2 ? 1 : 2;
  static_assert(K == 2 || K == 4, "candidates per record");
  static_assert(SETS >= 1 && SETS * 64 == PARTS * CPP, "whole sets");
  __shared__ unsigned short korig[64 * kWaves * kSlots];
  __shared__ SqRecords rec;
  __shared__ float pick_x[kCoopMaxPicks], pick_y[kCoopMaxPicks], pick_z[kCoopMaxPicks];
  __shared__ int pick_n, abort_flag;
  const int h = threadIdx.x, lane = h & 63;
  const int wave = __builtin_amdgcn_readfirstlane(h >> 6);
  // block -> (scene, part): the parts of a scene sit on block ids congruent mod 8 (one XCD under round-robin dispatch)
  const int t_ = blockIdx.x >> 3;
  const int part = t_ % PARTS;
  const int scene = (t_ / PARTS) * 8 + (blockIdx.x & 7);
  if (scene >= b) return;
  xyz += (size_t)scene * xyz_bstride;
  perm += (size_t)scene * n + (size_t)part * kPartPoints;
  idxs += (size_t)scene * idx_bstride;
  // exchange area of the scene: [8 * PARTS] round-0 words, then [2 parities][PARTS][CPP][5] candidate words
  exch += (size_t)scene * coop_multi_words(PARTS, K);
  unsigned long long *slots = exch + 8 * PARTS;
  unsigned short *korig_w = korig + (size_t)wave * 64 * SG;

  float px[SG], py[SG], pz[SG], pt[SG];
  float lox, loy, loz, hix, hiy, hiz;
  {
    float ax = 3.0e38f, ay = 3.0e38f, az = 3.0e38f, bx = -3.0e38f, by = -3.0e38f, bz = -3.0e38f;
#pragma unroll
    for (int j = 0; j < SG; ++j) {
      const int pos = (wave * 64 + lane) * SG + j;
      const int k = (int)perm[pos];
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
  if (h == 0) { pick_x[0] = xyz[0]; pick_y[0] = xyz[1]; pick_z[0] = xyz[2]; pick_n = 1; abort_flag = 0; }
  if (part == 0 && h == 0) idxs[0] = idx_add;             // the first pick is point 0 (sampling_gpu.cu:131-133)

  // Round 0 (wave 0): placement + visibility handshake for the optional workgroup-scope publishing stores, as in the
  // one-pick kernel above
  bool fast = false;
  if (wave == 0 && allow_fast) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;
    const bool reader0 = lane < PARTS;
    if (lane == 0) {
      __hip_atomic_store(exch + (size_t)part * 8 + 0, co_pack(xcc, 0x7fffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(exch + (size_t)part * 8 + 1, co_pack(0x5A5A0000u | (unsigned)part, 0x7ffffffeu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const unsigned long long *th = exch + (size_t)(reader0 ? lane : 0) * 8;
    bool seen = false;
    unsigned long long w0 = 0ull;
    for (int spins = 0; spins < 8192 && !seen; ++spins) {
      w0 = __hip_atomic_load(th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long w1 = __hip_atomic_load(th + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      seen = __ballot(reader0 && ((unsigned)(w0 >> 32) != 0x7fffffffu || (unsigned)(w1 >> 32) != 0x7ffffffeu)) == 0ull;
      if (!seen) __builtin_amdgcn_s_sleep(1);
    }
    fast = seen && __ballot(reader0 && (unsigned)w0 != xcc) == 0ull;    // (a part that does not see all tokens stays on agent scope)
  }
  __syncthreads();

  int r = 1;                                              // picks made so far
  for (unsigned round = 1;; ++round) {
    // ---- A. every wave: the picks of the last round against its points
    {
      const int np = pick_n;
      const float sx = pick_x[lane], sy = pick_y[lane], sz = pick_z[lane];
      const float gx = fmaxf(0.f, fmaxf(lox - sx, sx - hix));
      const float gy = fmaxf(0.f, fmaxf(loy - sy, sy - hiy));
      const float gz = fmaxf(0.f, fmaxf(loz - sz, sz - hiz));
      const float lb = d6_sqdist(gx, gy, gz);
      // (the first round always rescans: see fps_seq.hip)
      u64 need = __ballot(lane < np && (round == 1u || !(lb >= cmax)));
      if (need != 0ull) {
        while (need & (need - 1ull)) {
          const int i = __builtin_ctzll(need);
          need &= need - 1ull;
          sq_apply<SG>(d6_readlane_f(sx, i), d6_readlane_f(sy, i), d6_readlane_f(sz, i), px, py, pz, pt);
        }
        const int i = __builtin_ctzll(need);
        cmax = sq_rescan<SG, K>(d6_readlane_f(sx, i), d6_readlane_f(sy, i), d6_readlane_f(sz, i), log2s, px, py, pz, pt, korig_w, rec, wave,
                                np <= 1 ? min(K, 2) : K);       // (list depth: fps_seq.hip)
      }
    }
    if (r >= m || abort_flag) break;
    __syncthreads();                                      // the part's records are complete
    // ---- B. wave 0: publish, gather the scene's records, decide
    if (wave == 0) {
      unsigned long long *mine = slots + ((size_t)(round & 1) * PARTS + part) * CPP * 5;
      if (lane < CPP) {
        const int w = lane >> LOG2K, c = lane & (K - 1);
        const int o = w * kCandMax + c;
        // (slots c >= nc of a record have never been written in this launch if the wave's list has always been shorter: what
        // they hold is whatever the LDS held before the kernel started.  Readers ignore such a slot — once they know nc, so
        // nothing of the slot's own content may reach the nc field.)
        const unsigned kword = ((unsigned)rec.k[o] & 0xFFFFu) | ((unsigned)rec.nc[w] << 16);
        unsigned long long *d = mine + (size_t)lane * 5;
        const unsigned long long w0 = co_pack(__builtin_bit_cast(unsigned, rec.v[o]), round), w1 = co_pack(kword, round),
                                 w2 = co_pack(__builtin_bit_cast(unsigned, rec.x[o]), round),
                                 w3 = co_pack(__builtin_bit_cast(unsigned, rec.y[o]), round),
                                 w4 = co_pack(__builtin_bit_cast(unsigned, rec.z[o]), round);
        if (fast) {
          __hip_atomic_store(d + 0, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(d + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(d + 2, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(d + 3, w3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(d + 4, w4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          __hip_atomic_store(d + 0, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(d + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(d + 2, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(d + 3, w3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(d + 4, w4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // gather: global candidate 64 s + lane = candidate (64 s + lane) % CPP of part (64 s + lane) / CPP
      float cv[SETS], qx[SETS], qy[SETS], qz[SETS], bound_v[SETS];
      unsigned ntk[SETS];
      bool is_last[SETS];
      bool dead = false;
      const unsigned long long *base = slots + (size_t)(round & 1) * PARTS * CPP * 5;
#pragma unroll
      for (int s_ = 0; s_ < SETS; ++s_) {
        const unsigned long long *src = base + (size_t)(64 * s_ + lane) * 5;
        unsigned long long w0, w1, w2, w3, w4;
        int spins = 0;
        for (;;) {
          w0 = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w2 = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w3 = __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w4 = __hip_atomic_load(src + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool ok = (unsigned)(w0 >> 32) == round && (unsigned)(w1 >> 32) == round && (unsigned)(w2 >> 32) == round &&
                          (unsigned)(w3 >> 32) == round && (unsigned)(w4 >> 32) == round;
          if (__ballot(!ok) == 0ull) break;
          if (++spins > (1 << 21)) { dead = true; break; }   // ~2 s: a partner never arrived
          __builtin_amdgcn_s_sleep(1);
        }
        if (dead) break;
        const int nc = (int)(((unsigned)w1 >> 16) & 7u);
        const int slot = lane & (K - 1);
        cv[s_] = slot < nc ? __builtin_bit_cast(float, (unsigned)w0) : -1.0f;
        ntk[s_] = ~sq_tie_key((int)((unsigned)w1 & 0xFFFFu), log2s);      // ~tie key: larger wins
        qx[s_] = __builtin_bit_cast(float, (unsigned)w2);
        qy[s_] = __builtin_bit_cast(float, (unsigned)w3);
        qz[s_] = __builtin_bit_cast(float, (unsigned)w4);
        is_last[s_] = slot == nc - 1;
        bound_v[s_] = sq_group_max<K>(is_last[s_] ? cv[s_] : -2.0f);
      }
      int j = 0;
      if (dead) {
        if (lane == 0) { atomicExch(err, 1); abort_flag = 1; }
        if (part == 0) for (int i = r + lane; i < m; i += 64) idxs[i] = idx_add;   // in-range picks: nothing downstream may fault
      } else {
        // Keys and the per-candidate constants thr / alt: fps_seq.hip (an exact candidate's even key 2 bits(v) + 2 against a
        // region's odd unknown key 2 bits(bound) + 3; thr = the smallest key that makes the region exact through THIS
        // candidate, alt = what the lane puts forward otherwise: the unknown key from the last candidate's lane, nothing from
        // the others; an empty slot is never exact).  No reduction over a region's lanes is needed.
        // (alt = thr + 1 in the last candidate's lane: recomputed per pick from a value the compiler cannot see through — a
        // second array of per-round constants does not fit the 128 registers of a 1024-thread workgroup)
        unsigned thr[SETS];
#pragma unroll
        for (int s_ = 0; s_ < SETS; ++s_) {
          const unsigned ubk = ((__builtin_bit_cast(unsigned, bound_v[s_]) << 1) | 1u) + 2u;
          const unsigned ntk_last = sq_group_max_u32<K>(is_last[s_] ? ntk[s_] : 0u);
          thr[s_] = cv[s_] < 0.f ? 0xFFFFFFFFu : ntk[s_] >= ntk_last ? ubk - 1u : ubk;
        }
        const int jmax = min(max_picks, m - r);
        // (round 6, as in fps_seq.hip) picks are published after the loop: lane j remembers which lane AND which of its sets
        // pick j was (the candidate a lane puts forward changes from pick to pick here); coordinates and tie key are constants
        // of the round, fetched with cross-lane reads once per round
        int mws = 0;       // lane j: (set << 6) | lane of the candidate pick j was (ONE register: the kernel sits at its 128)
        bool go;       // ONE loop exit (fps_seq.hip: the decision that ends the round runs the rest of the body harmlessly)
        do {
          // this lane's best (key, ~tie key) over its sets as one 64-bit number
          u64 lbest = 0ull;
          int lset = 0;
#pragma unroll
          for (int s_ = 0; s_ < SETS; ++s_) {
            const unsigned ekey = (__builtin_bit_cast(unsigned, cv[s_]) << 1) + 2u;
            unsigned t = thr[s_];
            asm volatile("" : "+v"(t));
            const unsigned key = ekey >= t ? ekey : (is_last[s_] ? t + 1u : 0u);
            const u64 comp = ((u64)key << 32) | ntk[s_];
            const bool better = comp > lbest;
            lset = better ? s_ : lset;
            lbest = better ? comp : lbest;
          }
          const unsigned lkey = (unsigned)(lbest >> 32);
          const unsigned best = sq_wave_max_u32(lkey);
          go = (best & 1u) == 0u;                            // odd: an unknown region may hold the maximum: the round ends
          float ex = qx[0], ey = qy[0], ez = qz[0];
#pragma unroll
          for (int s_ = 1; s_ < SETS; ++s_) {
            const bool sel = lset == s_;
            ex = sel ? qx[s_] : ex; ey = sel ? qy[s_] : ey; ez = sel ? qz[s_] : ez;
          }
          const u64 tie = __ballot(lkey == best);
          int wl = __builtin_ctzll(tie);
          if (__popcll(tie) != 1) wl = sq_min_key_lane(tie, ~(unsigned)lbest);
          const float sx = d6_readlane_f(ex, wl), sy = d6_readlane_f(ey, wl), sz = d6_readlane_f(ez, wl);
          {
            int ls = lset;
            asm volatile("" : "+v"(ls));       // (a VGPR, whatever form the selects above left the set index in)
            mws = lane == j ? ((d6_readlane_i(ls, wl) << 6) | wl) : mws;
          }
          if constexpr (SETS % 2 == 0) {
            const sq_f32x2 c2x = {sx, sx}, c2y = {sy, sy}, c2z = {sz, sz};
#pragma unroll
            for (int s_ = 0; s_ < SETS; s_ += 2) {
              const sq_f32x2 dx = sq_f32x2{qx[s_], qx[s_ + 1]} - c2x;
              const sq_f32x2 dy = sq_f32x2{qy[s_], qy[s_ + 1]} - c2y;
              const sq_f32x2 dz = sq_f32x2{qz[s_], qz[s_ + 1]} - c2z;
              sq_f32x2 d = dy * dy;
              d = __builtin_elementwise_fma(dx, dx, d);
              d = __builtin_elementwise_fma(dz, dz, d);
              cv[s_] = d6_vmin(cv[s_], d[0]);
              cv[s_ + 1] = d6_vmin(cv[s_ + 1], d[1]);
            }
          } else {
#pragma unroll
            for (int s_ = 0; s_ < SETS; ++s_) cv[s_] = d6_vmin(cv[s_], d6_sqdist(qx[s_] - sx, qy[s_] - sy, qz[s_] - sz));
          }
          j += go ? 1 : 0;
        } while (go && j < jmax);
        {
          const int src = mws & 63, sset = mws >> 6;
          float mx = __shfl(qx[0], src), my = __shfl(qy[0], src), mz = __shfl(qz[0], src);
          unsigned mtk = (unsigned)__shfl((int)ntk[0], src);
#pragma unroll
          for (int s_ = 1; s_ < SETS; ++s_) {
            const float tx = __shfl(qx[s_], src), ty = __shfl(qy[s_], src), tz = __shfl(qz[s_], src);
            const unsigned tk = (unsigned)__shfl((int)ntk[s_], src);
            const bool sel = sset == s_;
            mx = sel ? tx : mx; my = sel ? ty : my; mz = sel ? tz : mz; mtk = sel ? tk : mtk;
          }
          if (lane < j) {
            pick_x[lane] = mx; pick_y[lane] = my; pick_z[lane] = mz;
            if (part == 0) idxs[r + lane] = sq_tie_key_point(~mtk, log2s) + idx_add;
          }
        }
      }
      if (lane == 0) pick_n = j;
    }
    __syncthreads();                                      // picks published
    r += pick_n;
  }
}

struct CoopLayout {
  size_t part_idx, perm, exch, err, total;
};

CoopLayout coop_layout(int b, int n) {
  CoopLayout L;
  const size_t items = (size_t)b * n;
  auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  L.err = off; off = align(off + 256);          // first: its offset must not depend on b (a workspace serves launches of fewer scenes)
  L.part_idx = off; off = align(off + items * 4);
  L.perm = off; off = align(off + items * 4);
  L.exch = off; off = align(off + (size_t)b * coop_multi_words(4, kMultiCands) * 8);    // (covers the one-pick kernel's 2 x 4 x 8 words too)
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
int det6d_fps_cell_sort_parts(int subscenes, int parts, int log2s, long long xyz_bstride, const float *xyz, const int *src, int *perm,
                              hipStream_t stream);      // fps_cells.hip

// CU count of the CURRENT device (cached per device: a process may drive several)
static int coop_device_cus() {
  static int cus_of[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cus_of[dev] <= 0) {
    hipDeviceProp_t prop;
    cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus_of[dev];
}
// A sampling launch covers scenes in multiples of 8 (the block -> scene map) with `parts` co-resident 1024-thread workgroups
// each: a device (or partition) with fewer than 8 x parts CUs cannot hold one such launch — its workgroups would poll partners
// that are never scheduled until the 2 s time-out — so det6d_fps_fused routes such a device to the memory-resident sampler
// (same picks) and det6d_fps_coop_launch refuses before anything is queued.
bool det6d_fps_coop_fits_device(int n) {
  return det6d_fps_coop_handles(n) && coop_device_cus() >= 8 * (n / kPartPoints);
}

int det6d_fps_coop_launch(int b, int n, int m, int log2s, long long xyz_bstride, long long idx_bstride, int idx_add,
                          const float *xyz, void *workspace, int *idx, hipStream_t stream) {
  if (!det6d_fps_coop_handles(n) || b <= 0 || b > 4096 || !workspace || ((uintptr_t)workspace & 255)) return DET6D_EINVAL;
  const CoopLayout L = coop_layout(b, n);
  char *ws = (char *)workspace;
  int *part_idx = (int *)(ws + L.part_idx);
  unsigned *vals_out = (unsigned *)(ws + L.perm);
  unsigned long long *exch = (unsigned long long *)(ws + L.exch);
  int *err = (int *)(ws + L.err);
  const int parts = n / kPartPoints;
  if (!det6d_fps_coop_fits_device(n)) {
    det6d_set_error("det6d_fps (cooperative): the device has fewer than 8 x parts compute units", hipErrorInvalidConfiguration);
    return DET6D_EINVAL;
  }
  const int cus = coop_device_cus();
  static const int multi = det6d_env_int("DET6D_FPS_COOP_MULTI", 1);      // experiments build: 0 = the one-pick kernel
  const int exch_words = multi ? (int)coop_multi_words(parts, kMultiCands) : 2 * parts * kSlotWords;
  // pre-pass: spatial parts of 16384 points, then the single-workgroup sampler's k-d order inside every part
  if (parts == 4) hipLaunchKernelGGL(coop_split_kernel<4>, dim3(b), dim3(1024), 0, stream, n, xyz_bstride, xyz, part_idx, exch, exch_words);
  else hipLaunchKernelGGL(coop_split_kernel<2>, dim3(b), dim3(1024), 0, stream, n, xyz_bstride, xyz, part_idx, exch, exch_words);
  const int sort_rc = det6d_fps_cell_sort_parts(b * parts, parts, log2s, xyz_bstride, xyz, part_idx, (int *)vals_out, stream);
  if (sort_rc != DET6D_OK) return sort_rc;
  // The parts of a scene poll each other, so all workgroups of a launch must be resident together: ONE launch never asks for
  // more than one 1024-thread workgroup per CU — more scenes are sampled chunk by chunk on the stream (launches of a stream do
  // not overlap), so that a single call cannot starve itself whatever `b` is.  (Two calls in flight on DIFFERENT streams are
  // still the caller's to bound: det6d_ops.h, ScenePipeline does it.)
  const int chunk = cus / parts / 8 * 8;      // scenes per sampling launch (a multiple of 8: the block -> scene map); >= 8, checked above
  // DET6D_FPS_COOP_FAST=1: workgroup-scope publishing stores where the placement test and the handshake allow (see the top)
  static const int allow_fast = det6d_switch_int("DET6D_FPS_COOP_FAST", 0) ? 1 : 0;
  // (clamped like fps_seq.hip's: 0 would never advance a round, more than kCoopMaxPicks would write past the pick arrays in LDS)
  static const int max_picks = std::min(std::max(det6d_env_int("DET6D_FPS_SEQ_PICKS", kCoopMaxPicks), 1), (int)kCoopMaxPicks);
#ifdef DET6D_EXPERIMENTS
  det6d_dbg_poison_lds_hook(stream);      // DET6D_DBG_POISON_LDS: fps_seq.hip
#endif
  for (int s0 = 0; s0 < b; s0 += chunk) {
    const int bc = std::min(chunk, b - s0);
    const int grid = 8 * parts * ((bc + 7) / 8);
    const float *x0 = xyz + (size_t)s0 * xyz_bstride;
    const unsigned *perm0 = vals_out + (size_t)s0 * n;
    int *idx0 = idx + (size_t)s0 * idx_bstride;
    unsigned long long *exch0 = exch + (size_t)s0 * exch_words;
    if (multi) {
      if (parts == 4)
        hipLaunchKernelGGL((fps_coop_multi_kernel<4, kMultiCands>), dim3(grid), dim3(1024), 0, stream, bc, n, m, log2s, xyz_bstride, idx_bstride,
                           idx_add, x0, perm0, idx0, exch0, err, allow_fast, max_picks);
      else
        hipLaunchKernelGGL((fps_coop_multi_kernel<2, kMultiCands>), dim3(grid), dim3(1024), 0, stream, bc, n, m, log2s, xyz_bstride, idx_bstride,
                           idx_add, x0, perm0, idx0, exch0, err, allow_fast, max_picks);
      continue;
    }
#ifdef DET6D_EXPERIMENTS
    if (parts == 4)
      hipLaunchKernelGGL(fps_coop_kernel<4>, dim3(grid), dim3(1024), 0, stream, bc, n, m, log2s, xyz_bstride, idx_bstride, idx_add, x0,
                         perm0, idx0, exch0, err, allow_fast);
    else
      hipLaunchKernelGGL(fps_coop_kernel<2>, dim3(grid), dim3(1024), 0, stream, bc, n, m, log2s, xyz_bstride, idx_bstride, idx_add, x0,
                         perm0, idx0, exch0, err, allow_fast);
#else
    return DET6D_EINVAL;
#endif
  }
  return det6d_check_launch(multi ? "det6d_fps (cooperative, multi-pick)" : "det6d_fps (cooperative)");
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
