// mlp_group.hip — a WIDE grouped MLP (the SA3 / head-SA radius groups: [.. -> 128|256 -> 128|256|512 -> 256|512|1024])
// as ONE launch: first layer from the per-point partial sums (expand.hip's arithmetic), second and third layer as fp32
// MFMA GEMMs, class / nsample max-pool, all on a 32-row tile whose activations never leave the CU.
//
// Replaces, per radius group, det6d_group_expand + two det6d_linear launches and their (rows x C1) and (rows x C2) fp32
// intermediates in memory (pointnet2_modules.py:462-494: three Conv2d/BN/ReLU, mask, max_pool2d).
//
// Design for gfx950 (why it is not the 128 x 64 tile kernel of linear.hip run three times):
//  * a workgroup = 4 waves owns 32 rows through all layers; wave w owns the quarter [w * C/4, (w+1) * C/4) of every
//    layer's output columns (C/128 accumulator tiles of 32 x 32: 8 for the 1024-wide layer = 128 accumulator VGPRs);
//  * activations live in LDS, row-major with an ODD row stride (C + 1 floats): the MFMA A fragment (row = lane & 31,
//    k = 2s + (lane >> 5)) is one conflict-free ds_read_b32, shared by all the wave's column tiles;
//  * weights are NOT staged: with one 32-row tile no two waves share a weight element, so every wave reads its B
//    fragments (k = 2s + (lane >> 5), 32 consecutive columns = one 128-byte line per lane half) straight from L2 with
//    buffer_load_dwordx4 (each lane owns 4 consecutive columns = the operands of 4 accumulator tiles), two blocks of 32 MFMAs
//    (2 x 2048 matrix cycles) ahead of their use, per-lane offset computed once and
//    the k / column offsets as scalars: the K loops contain no barrier, no LDS store and no vector-ALU instruction
//    (on gfx950 the fp32 MFMA shares the vector ALU: DESIGN.md §8);
//  * two workgroup barriers per tile (after the first layer is in LDS, after the second), none inside a layer.
// Arithmetic: every output is the oracle's chain (features ascending, then dx, dy, dz for layer 1; k ascending for
// layers 2, 3; + shift; ReLU; max over the rows of a centre) — bit-identical to the three-launch path.
#include "common.h"

namespace {

D6_GEMM_PRIO_DECL

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4g __attribute__((ext_vector_type(4)));

struct GroupArgs {
  int rows;                                       // dense: b * m * ns; compact: capacity of the row list
  const float *p; int ldp; int pcol0;             // per-point partial sums of layer 1 (this group's columns at pcol0)
  const float *w1; int ldw1; const float *s1;     // rows 0..2 of w1: the coordinate rows
  const float *w2; int ldw2; const float *s2;
  const float *w3; int ldw3; const float *s3;
  const float *pts; int ldpts;
  const float *ctr; int ldctr;
  const int *idx; int n, m, ns; const int *cnt;   // dense rows
  const int *hdr; const int *crow_p; const int *crow_c;   // compact rows
  float *y; int ldy; int col0;
  int pre;                                        // A/B switch: list entries of the next tile requested a K loop ahead
  int yvec;                                       // y rows are 16-byte aligned (ldy, col0 multiples of 4): vector stores allowed
  int *ticket;                                    // compact rows: tile ticket + exit counter (hdr[10], hdr[11]); nullptr: static tiles
};

// ---- tiles by ticket (round 5) --------------------------------------------------------------------------------------------
// The group kernels are persistent: one workgroup per CU (or 2-3) walks the 32-row tiles of the launch.  With a STATIC walk
// (tile += gridDim.x) the launch ends with its slowest workgroup, and in the pipeline the CUs are not alike: ~30 of the 256
// host a sampler workgroup (16 waves at raised priority for 2 ms), others a ball-query or list-builder wave — the wide head
// group ran 1.5-1.6 ms per launch under load against 0.98 ms alone.  Now a workgroup takes its FIRST tile by its block index
// and every further one from a ticket counter in the list header: fast CUs take more tiles.  The ticket of the next tile is
// drawn at the top of the current one (its latency hides behind the gather), published through LDS by tile parity, so the
// next tile's list entries can still be requested a K loop ahead.  The counter pair cleans itself: the last workgroup to leave
// (exit counter == workgroups that had a tile) zeroes both, so a launch may be replayed without rebuilding the list
// (bench: family_saturated); compact_place_kernel zeroes them too when it builds the list.  Results do not depend on which
// workgroup computes a tile (a tile's outputs are a function of its rows; multi-part centres combine by an order-independent
// integer max).
__device__ __forceinline__ int g_draw_ticket(int *ticket, int grid) {
  return grid + __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void g_leave(int *ticket, int workers) {
  // (this thread's last ticket draw has returned — its value was used — before this add is issued)
  const int gone = __hip_atomic_fetch_add(ticket + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (gone == workers - 1) {
    __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ticket + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- compact-row helpers (same conventions as mlp_chain.hip / linear.hip; see compact.hip for the list layout) ----
__device__ __forceinline__ int g_class(int row0, int h1, int h2, int h3, int h4, int h5) {
  return row0 < h1 ? 32 : row0 < h2 ? 16 : row0 < h3 ? 8 : row0 < h4 ? 4 : row0 < h5 ? 2 : 1;
}
__device__ __forceinline__ int g_out_row(int s, int qq, int kh) {
  if (s < 4) return -1;
  if (s == 4) return 8 * qq + 4 * kh;
  if (kh) return -1;
  if (s == 8) return 8 * qq;
  if (s == 16) return (qq & 1) ? -1 : 8 * qq;
  return qq == 0 ? 0 : -1;
}
__device__ __forceinline__ void g_store(float *dst, float val, int tag) {
  if (tag & 0x20000000) __hip_atomic_fetch_max(reinterpret_cast<int *>(dst), __builtin_bit_cast(int, val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *dst = val;
}
__device__ __forceinline__ void g_pool(float (&v)[4], int s) {
  if (s == 4) return;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) v[qq] = d6_vmax(v[qq], __shfl_xor(v[qq], 32));
  if (s == 16) {
    v[0] = d6_vmax(v[0], v[1]);
    v[2] = d6_vmax(v[2], v[3]);
  } else if (s == 32) {
    v[0] = d6_vmax(d6_vmax(v[0], v[1]), d6_vmax(v[2], v[3]));
  }
}

// Column owned by (accumulator tile j, lane l31) inside a wave's C/4-wide quarter: VW = min(TN, 4) consecutive columns
// per lane so that ONE buffer_load_dwordx{VW} brings the B operands of VW tiles (any assignment of columns to
// (tile, lane) is a valid GEMM; the epilogues use the same map).
template <int TN>
__device__ __forceinline__ int tile_col(int j, int l31) {
  constexpr int VW = TN >= 4 ? 4 : TN;
  return (j / VW) * (32 * VW) + VW * l31 + (j % VW);
}

template <int VW> struct BVec;
template <> struct BVec<1> { typedef float T; };
template <> struct BVec<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct BVec<4> { typedef float T __attribute__((ext_vector_type(4))); };

// Stores / max-combines the VW values a lane holds for ONE pooled row: VW CONSECUTIVE columns (4 l31 + t inside a group of
// 32 VW columns: the column map of tile_col, which follows from the 16-byte B-fragment loads).
//  * single-part centres: one 8- / 16-byte store per lane, the half-wave covers the group's 32 VW columns contiguously
//    (round 2: VW dword stores per lane, each instruction touching every VW-th dword of the span);
//  * multi-part centres (integer atomic max): the values go through the wave's LDS scratch so that every atomic instruction
//    covers 32 CONTIGUOUS columns = two 64-byte memory-side requests instead of 2 VW partly used ones (the atomics of this
//    kernel were 499 of the pass's 786 MB of memory-side writes: profiles/r02_zm_pmc_summary.json).
// Must be called by all 64 lanes (tag < 0: nothing to store for this lane's half); tags are uniform per half.
template <int VW>
__device__ __forceinline__ void g_store_group(float *rowptr, const float (&v)[VW], const int tag, float *__restrict__ scr,
                                              const int lane, const int yvec) {
  const bool live = tag >= 0;
  const int l31 = lane & 31;
  if constexpr (VW == 1) {
    if (live) g_store(rowptr + l31, v[0], tag);
  } else {
    if (!yvec) {
#pragma unroll
      for (int t = 0; t < VW; ++t)
        if (live) g_store(rowptr + VW * l31 + t, v[t], tag);
      return;
    }
    const bool atomic = live && (tag & 0x20000000);
    typename BVec<VW>::T pack;
#pragma unroll
    for (int t = 0; t < VW; ++t) pack[t] = v[t];
    if (live && !atomic) *reinterpret_cast<typename BVec<VW>::T *>(rowptr + VW * l31) = pack;
    if (__ballot(atomic) != 0ull) {                       // wave-uniform
      *reinterpret_cast<typename BVec<VW>::T *>(scr + lane * VW) = pack;
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
      const float *half = scr + (lane & 32) * VW;         // this half's 32 VW values, column order
      float x[VW];
#pragma unroll
      for (int t = 0; t < VW; ++t) x[t] = half[32 * t + l31];
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();                    // the scratch may be rewritten by the next call
      if (atomic) {
#pragma unroll
        for (int t = 0; t < VW; ++t)
          __hip_atomic_fetch_max(reinterpret_cast<int *>(rowptr + 32 * t + l31), __builtin_bit_cast(int, x[t]), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

template <int VW>
__device__ __forceinline__ typename BVec<VW>::T load_b(const __amdgpu_buffer_rsrc_t srd, uint32_t voff, int soff) {
  if constexpr (VW == 4) return __builtin_bit_cast(typename BVec<4>::T, __builtin_amdgcn_raw_buffer_load_b128(srd, voff, soff, 0));
  else if constexpr (VW == 2) return __builtin_bit_cast(typename BVec<2>::T, __builtin_amdgcn_raw_buffer_load_b64(srd, voff, soff, 0));
  else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, voff, soff, 0));
}
template <int VW>
__device__ __forceinline__ float bget(const typename BVec<VW>::T &v, int i) {
  if constexpr (VW == 1) return v;
  else return v[i];
}

// One layer of the tile: acc[j] += X (32 x K, LDS, row stride LDX) x W[:, this wave's columns].  The B fragments come
// straight from memory, a block (UK k-steps x TN tiles = 32 MFMAs = 2048 matrix cycles) at a time, TWO blocks ahead of
// their use (L2 latency under load exceeds one block), with one load per VW tiles.
#ifdef DET6D_EXPERIMENTS
// timing experiments only (DET6D_GROUP_WHATIF=1, wrong results): every B fragment comes from the first two weight rows, i.e.
// from the CU's own cache: how much of the kernel's time is the L2 -> CU weight stream?
__device__ int d6_group_kmul = 1;
#define D6_KOFF(x) (d6_group_kmul * (x))
// phase timers of mlp_group_kernel<256, 512, 1024> (100 MHz wall clock as seen by wave 0 of every workgroup, summed):
// [0] layer 1 incl. its barrier, [1] layer 2 K loop, [2] layer 2 epilogue + barrier, [3] layer 3 K loop, [4] pooling / stores,
// [5] tiles, [6] whole kernel, [7] workgroups
__device__ int d6_group_phase_c2 = 512, d6_group_phase_c3 = 1024;   // which instantiation is timed (DET6D_GROUP_PHASE_C2 / _C3)
__device__ unsigned long long d6_group_phase[8 + 18];   // [8] shader-clock cycles of the kernel (wave 0), [9] .. K-loop wall time of layer 3 by wave (8) and of layer 2 by wave (8), [25] unused
#define D6_PHASE_DECL const unsigned long long ph_c0 = clock64(); unsigned long long ph_w3 = 0, ph_w2 = 0, ph_wt = 0; unsigned long long ph_t = wall_clock64(), ph_acc[5] = {0, 0, 0, 0, 0}, ph_tiles = 0; const unsigned long long ph_start = ph_t; const bool ph_on = C3 == d6_group_phase_c3 && C2 == d6_group_phase_c2;
#define D6_PHASE(i) do { if (ph_on) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = wall_clock64(); ph_acc[i] += now_ - ph_t; ph_t = now_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#define D6_WAVE_T0 do { if (ph_on) { __builtin_amdgcn_sched_barrier(0); ph_wt = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define D6_WAVE_T1(x) do { if (ph_on) { __builtin_amdgcn_sched_barrier(0); x += wall_clock64() - ph_wt; __builtin_amdgcn_sched_barrier(0); } } while (0)
#define D6_PHASE_END do { if (ph_on && tid == 0) { for (int i_ = 0; i_ < 5; ++i_) atomicAdd(&d6_group_phase[i_], ph_acc[i_]); atomicAdd(&d6_group_phase[5], ph_tiles); atomicAdd(&d6_group_phase[6], wall_clock64() - ph_start); atomicAdd(&d6_group_phase[7], 1ull); atomicAdd(&d6_group_phase[8], clock64() - ph_c0); } if (ph_on && (tid & 63) == 0) { atomicAdd(&d6_group_phase[9 + (tid >> 6)], ph_w3); atomicAdd(&d6_group_phase[17 + (tid >> 6)], ph_w2); } } while (0)
#else
#define D6_PHASE_DECL
#define D6_PHASE(i)
#define D6_WAVE_T0
#define D6_WAVE_T1(x)
#define D6_PHASE_END
#define D6_KOFF(x) (x)
#endif

template <int K, int TN, int LDX>
struct GroupLayer {
  static constexpr int VW = TN >= 4 ? 4 : TN;
  static constexpr int NV = TN / VW;         // loads per k-step
  static constexpr int KS = K / 2;           // k-steps of two
  static constexpr int UK = 32 / TN;         // k-steps per block
  static constexpr int NB = KS / UK;         // blocks
  static_assert(KS % UK == 0 && NB >= 2, "block structure");
  typedef typename BVec<VW>::T bvec;
  bvec b0[UK][NV], b1[UK][NV], b2[UK][NV];   // three register sets in rotation: while block i computes, blocks i+1 and i+2 are in flight

  __device__ __forceinline__ static void fetch(bvec (&b)[UK][NV], const __amdgpu_buffer_rsrc_t srd, const uint32_t voff,
                                               const int ldw_bytes, const int blk) {
#pragma unroll
    for (int u = 0; u < UK; ++u)
#pragma unroll
      for (int v = 0; v < NV; ++v) b[u][v] = load_b<VW>(srd, voff, D6_KOFF(2 * (blk * UK + u) * ldw_bytes) + 128 * VW * v);
  }
  __device__ __forceinline__ static void compute(const bvec (&b)[UK][NV], const float *__restrict__ X, f32x16 (&acc)[TN],
                                                 const int l31, const int kh, const int blk) {
    float a[UK];
    const float *xa = X + l31 * LDX + 2 * blk * UK + kh;
#pragma unroll
    for (int u = 0; u < UK; ++u) a[u] = xa[2 * u];
#pragma unroll
    for (int u = 0; u < UK; ++u)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bget<VW>(b[u][j / VW], j % VW), acc[j], 0, 0, 0);
  }
  // the first two blocks of weights: issued BEFORE the phase that produces the layer's input (the first layer's gather,
  // the second layer's epilogue and the barrier behind it), so that the K loop does not start with an L2 round trip
  __device__ __forceinline__ void start(const __amdgpu_buffer_rsrc_t srd, const uint32_t voff, const int ldw_bytes) {
    fetch(b0, srd, voff, ldw_bytes, 0);
    fetch(b1, srd, voff, ldw_bytes, 1);
  }
  // acc[j] += X (32 x K, LDS) x W[:, this wave's columns].  Every fetch is unconditional (past the end it re-reads the
  // last block): behind a conditional fetch the compiler's s_waitcnt vmcnt accounting assumes the shorter queue and drains
  // the ring.
  __device__ __forceinline__ void run(const float *__restrict__ X, const __amdgpu_buffer_rsrc_t srd, const uint32_t voff,
                                      const int ldw_bytes, f32x16 (&acc)[TN], const int l31, const int kh) {
    auto clamp = [](int blk) { return blk < NB ? blk : NB - 1; };
    int blk = 0;
#pragma unroll 1
    for (; blk + 3 <= NB; blk += 3) {
      fetch(b2, srd, voff, ldw_bytes, blk + 2);
      compute(b0, X, acc, l31, kh, blk);
      fetch(b0, srd, voff, ldw_bytes, clamp(blk + 3));
      compute(b1, X, acc, l31, kh, blk + 1);
      fetch(b1, srd, voff, ldw_bytes, clamp(blk + 4));
      compute(b2, X, acc, l31, kh, blk + 2);
    }
    if (blk < NB) {             // NB mod 3 == 1 or 2 (NB is a power of two)
      compute(b0, X, acc, l31, kh, blk);
      if (blk + 1 < NB) compute(b1, X, acc, l31, kh, blk + 1);
    }
  }
};

// ---- layer 1 of a 32-row tile (expand.hip's arithmetic): EPR threads per row, thread (row = tid / EPR, q = tid % EPR)
// produces columns 4q + 4 EPR i of its row:
// X1[row][c] = relu(fma(dz, W1[2][c], fma(dy, W1[1][c], fma(dx, W1[0][c], P[p][c]))) + s1[c]) ----
// The row's list entry (compact: tag + point row; dense: neighbour index) is the first of two dependent global loads of the
// gather; group_row_entry() requests it for the NEXT tile before the current tile's third layer (ahead by a whole K loop).
template <bool COMPACT, int EPR>
__device__ __forceinline__ void group_row_entry(const GroupArgs &g, const int tile, const int tid, int &e0, int &e1) {
  const int r = tile * 32 + tid / EPR;
  if (COMPACT) { e0 = g.crow_c[r]; e1 = g.crow_p[r]; }
  else { e0 = g.idx[r]; e1 = 0; }
}

template <int C1, bool COMPACT, int EPR>
__device__ __forceinline__ void group_layer1(const GroupArgs &g, const int tile, const int tid, float *__restrict__ X1,
                                             int *__restrict__ tagbuf, const int e0, const int e1) {
  constexpr int LD1 = C1 + 1;
  const int erow = tid / EPR, eq = tid % EPR;
    {
      const int r = tile * 32 + erow;
      long long prow;
      int cj;
      bool real = true;
      if (COMPACT) {
        const int tag = e0;
        if (eq == 0) tagbuf[erow] = tag;     // the pooling epilogue takes the rows' tags from LDS (visible after the barrier)
        real = tag >= 0;
        cj = tag & 0x1fffffff;
        prow = e1;
      } else {
        cj = r / g.ns;
        prow = (long long)(cj / g.m) * g.n + e0;
      }
      float dx = 0.f, dy = 0.f, dz = 0.f;
      if (real) {
        const float *pt = g.pts + prow * g.ldpts;
        const float *ce = g.ctr + (long long)cj * g.ldctr;
        dx = pt[0] - ce[0]; dy = pt[1] - ce[1]; dz = pt[2] - ce[2];
      }
      const float *prow_p = g.p + (real ? prow : 0) * g.ldp + g.pcol0;
      float *xr = X1 + erow * LD1;
#pragma unroll 4
      for (int c = 4 * eq; c < C1; c += 4 * EPR) {
        const f32x4g pv = *reinterpret_cast<const f32x4g *>(prow_p + c);
        const f32x4g wx = *reinterpret_cast<const f32x4g *>(g.w1 + c);
        const f32x4g wy = *reinterpret_cast<const f32x4g *>(g.w1 + g.ldw1 + c);
        const f32x4g wz = *reinterpret_cast<const f32x4g *>(g.w1 + 2 * g.ldw1 + c);
        const f32x4g sh = *reinterpret_cast<const f32x4g *>(g.s1 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = d6_relu(D6_FMA(dz, wz[e], D6_FMA(dy, wy[e], D6_FMA(dx, wx[e], pv[e]))) + sh[e]);
          xr[c + e] = real ? v : 0.f;
        }
      }
    }
}

// ---- max-pool of the third layer's accumulators over the rows of a centre + shift + ReLU + store (compact rows: class
// layout of compact.hip, parts of a centre combined with an atomic max; dense rows: nsample 32 / 16 per centre) ----
template <int TN3, bool COMPACT>
__device__ __forceinline__ void group_pool_store(const GroupArgs &g, const int tile, f32x16 (&acc)[TN3], const float (&sh3)[TN3],
                                                 const int colbase, const int l31, const int kh, const int h1, const int h2,
                                                 const int h3, const int h4, const int h5, const int *__restrict__ tagbuf,
                                                 float *__restrict__ scr) {
    constexpr int VW = TN3 >= 4 ? 4 : TN3;            // consecutive columns per lane (tile_col)
    constexpr int NG = TN3 / VW;                      // groups of 32 VW columns
    const int lane = l31 + 32 * kh;
    if (COMPACT) {
      const int sc = g_class(tile * 32, h1, h2, h3, h4, h5);
      if (sc < 4) {       // classes 1, 2: every accumulator (pair) is a centre part of its own
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (sc == 2 && (e & 1)) continue;
          const int tag = tagbuf[(e & 3) + 8 * (e >> 2) + 4 * kh];
          float *dst = g.y + (size_t)(tag < 0 ? 0 : tag & 0x1fffffff) * g.ldy + g.col0 + colbase;
#pragma unroll
          for (int jg = 0; jg < NG; ++jg) {
            float v[VW];
#pragma unroll
            for (int t = 0; t < VW; ++t) {
              const int j = jg * VW + t;
              const float raw = sc == 2 ? d6_vmax(acc[j][e], acc[j][e + 1 < 16 ? e + 1 : e]) : acc[j][e];
              v[t] = (tag & 0x40000000) ? 0.f : d6_relu(raw + sh3[j]);
            }
            g_store_group<VW>(dst + jg * 32 * VW, v, tag, scr, lane, g.yvec);
          }
        }
      } else {
        int oc[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int rr = g_out_row(sc, qq, kh);
          oc[qq] = rr >= 0 ? tagbuf[rr] : -1;
        }
#pragma unroll
        for (int jg = 0; jg < NG; ++jg) {
          float q[VW][4];
#pragma unroll
          for (int t = 0; t < VW; ++t) {
            const int j = jg * VW + t;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
              q[t][qq] = d6_vmax(d6_vmax(acc[j][4 * qq], acc[j][4 * qq + 1]), d6_vmax(acc[j][4 * qq + 2], acc[j][4 * qq + 3]));
            g_pool(q[t], sc);
          }
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            float v[VW];
#pragma unroll
            for (int t = 0; t < VW; ++t) v[t] = (oc[qq] & 0x40000000) ? 0.f : d6_relu(q[t][qq] + sh3[jg * VW + t]);
            float *dst = g.y + (size_t)(oc[qq] < 0 ? 0 : oc[qq] & 0x1fffffff) * g.ldy + g.col0 + colbase + jg * 32 * VW;
            g_store_group<VW>(dst, v, oc[qq], scr, lane, g.yvec);
          }
        }
      }
    } else {
      // dense rows: a tile is one centre (nsample 32) or two (nsample 16); empty balls pool to 0
      const int c0 = g.ns == 32 ? tile : 2 * tile;
      const int cnt0 = g.cnt[c0], cnt1 = g.ns == 32 ? 0 : g.cnt[c0 + 1];
      const int plain = kh == 0 ? 0 : -1;              // lanes of half 0 store (tag 0: a plain store), half 1 holds the same values
#pragma unroll
      for (int jg = 0; jg < NG; ++jg) {
        float v0[VW], v1[VW];
#pragma unroll
        for (int t = 0; t < VW; ++t) {
          const int j = jg * VW + t;
          float q[4];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const float mq = d6_vmax(d6_vmax(acc[j][4 * qq], acc[j][4 * qq + 1]), d6_vmax(acc[j][4 * qq + 2], acc[j][4 * qq + 3]));
            q[qq] = d6_vmax(mq, __shfl_xor(mq, 32));
          }
          if (g.ns == 32) {
            const float mx = d6_relu(d6_vmax(d6_vmax(q[0], q[1]), d6_vmax(q[2], q[3])) + sh3[j]);
            v0[t] = cnt0 > 0 ? mx : 0.f;
            v1[t] = 0.f;
          } else {
            v0[t] = cnt0 > 0 ? d6_relu(d6_vmax(q[0], q[1]) + sh3[j]) : 0.f;
            v1[t] = cnt1 > 0 ? d6_relu(d6_vmax(q[2], q[3]) + sh3[j]) : 0.f;
          }
        }
        float *dst = g.y + g.col0 + colbase + jg * 32 * VW;
        g_store_group<VW>(dst + (size_t)c0 * g.ldy, v0, plain, scr, lane, g.yvec);
        if (g.ns != 32) g_store_group<VW>(dst + (size_t)(c0 + 1) * g.ldy, v1, plain, scr, lane, g.yvec);
      }
    }
}

template <int C1, int C2, int C3, bool COMPACT, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 4 && C3 <= 256) ? 3 : 1) void mlp_group_kernel(const GroupArgs g) {
  D6_GEMM_PRIO_APPLY();
  constexpr int LD1 = C1 + 1, LD2 = C2 + 1;
  constexpr int TN2 = C2 / (32 * NW), TN3 = C3 / (32 * NW);     // accumulator tiles per wave: a wave owns 1 / NW of every layer's columns
  static_assert(TN2 >= 1 && TN3 >= 1, "every wave needs at least one 32-column tile per layer");
  extern __shared__ float lds[];
  float *X1 = lds;
  float *X2 = lds + 32 * LD1;
  int *tags = reinterpret_cast<int *>(lds + 32 * (LD1 + LD2));    // 2 x 32 row tags, by tile parity (no barrier between a tile's
  int it = 0;                                                     // epilogue and the next tile's first layer)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  float *scr = reinterpret_cast<float *>(tags + 64) + wave * (64 * (TN3 >= 4 ? 4 : TN3));   // wave-private: g_store_group
  const int live_tiles = (COMPACT ? g.hdr[0] : g.rows) / 32;
  if ((int)blockIdx.x >= live_tiles) return;
  int h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0;
  if (COMPACT) { h1 = g.hdr[1]; h2 = g.hdr[2]; h3 = g.hdr[3]; h4 = g.hdr[4]; h5 = g.hdr[5]; }

  const __amdgpu_buffer_rsrc_t srd2 = __builtin_amdgcn_make_buffer_rsrc((void *)g.w2, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd3 = __builtin_amdgcn_make_buffer_rsrc((void *)g.w3, 0, 0xffffffff, 0x00020000);
  const uint32_t voff2 = (uint32_t)(kh * g.ldw2 + wave * (C2 / NW) + tile_col<TN2>(0, l31)) * 4u;
  const uint32_t voff3 = (uint32_t)(kh * g.ldw3 + wave * (C3 / NW) + tile_col<TN3>(0, l31)) * 4u;
  float sh2[TN2], sh3[TN3];
#pragma unroll
  for (int j = 0; j < TN2; ++j) sh2[j] = g.s2[wave * (C2 / NW) + tile_col<TN2>(j, l31)];
#pragma unroll
  for (int j = 0; j < TN3; ++j) sh3[j] = g.s3[wave * (C3 / NW) + tile_col<TN3>(j, l31)];

  D6_PHASE_DECL
  int e0, e1;
  group_row_entry<COMPACT, 2 * NW>(g, blockIdx.x, tid, e0, e1);
  __shared__ int next_tile_s[2];
  int *const ticket = COMPACT ? g.ticket : nullptr;
  for (int tile = blockIdx.x; tile < live_tiles;) {
#ifdef DET6D_EXPERIMENTS
    ++ph_tiles;
#endif
    int drawn = 0;
    if (ticket && tid == 0) drawn = g_draw_ticket(ticket, (int)gridDim.x);
    GroupLayer<C1, TN2, LD1> second;
    GroupLayer<C2, TN3, LD2> third;
    second.start(srd2, voff2, g.ldw2 * 4);
    const int par = it & 1;
    int *tagbuf = tags + 32 * (it++ & 1);
    if (!g.pre) group_row_entry<COMPACT, 2 * NW>(g, tile, tid, e0, e1);
    group_layer1<C1, COMPACT, 2 * NW>(g, tile, tid, X1, tagbuf, e0, e1);
    if (ticket && tid == 0) next_tile_s[par] = drawn;
    __syncthreads();
    const int next_tile = ticket ? next_tile_s[par] : tile + (int)gridDim.x;
    D6_PHASE(0);
    // ---- layer 2: X2 = relu(X1 W2 + s2) ----
    {
      f32x16 acc[TN2];
#pragma unroll
      for (int j = 0; j < TN2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
      D6_WAVE_T0;
      second.run(X1, srd2, voff2, g.ldw2 * 4, acc, l31, kh);
      third.start(srd3, voff3, g.ldw3 * 4);
      D6_WAVE_T1(ph_w2);
      D6_PHASE(1);
#pragma unroll
      for (int j = 0; j < TN2; ++j) {
        float *xc = X2 + wave * (C2 / NW) + tile_col<TN2>(j, l31);
#pragma unroll
        for (int e = 0; e < 16; ++e) xc[((e & 3) + 8 * (e >> 2) + 4 * kh) * LD2] = d6_relu(acc[j][e] + sh2[j]);
      }
    }
    __syncthreads();
    D6_PHASE(2);
    // ---- layer 3 + pooling ----
    f32x16 acc[TN3];
#pragma unroll
    for (int j = 0; j < TN3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    D6_WAVE_T0;
    if (g.pre) {          // the next tile's list entries, a K loop ahead of its first layer (the last tile re-reads its own)
      const int nt = next_tile;
      group_row_entry<COMPACT, 2 * NW>(g, nt < live_tiles ? nt : tile, tid, e0, e1);
    }
    third.run(X2, srd3, voff3, g.ldw3 * 4, acc, l31, kh);
    D6_WAVE_T1(ph_w3);
    D6_PHASE(3);
    group_pool_store<TN3, COMPACT>(g, tile, acc, sh3, wave * (C3 / NW), l31, kh, h1, h2, h3, h4, h5, tagbuf, scr);
    // no barrier here: the next tile's layer 1 writes X1, which every wave finished reading before the barrier above;
    // X2 is rewritten only after the next tile's first barrier, which no wave passes before it has left layer 3
    D6_PHASE(4);
    tile = next_tile;
  }
  if (ticket && tid == 0) g_leave(ticket, min((int)gridDim.x, live_tiles));
  D6_PHASE_END;
}


// ------------------------------------------------------------------------------------------------------------------------
// Streaming form for the widest groups (C2 >= 256): the second layer is produced 128 columns at a time and consumed at once
// as a K-chunk of the third layer, so a tile keeps 32 x (C1 + 2 x 128) floats in LDS (66 KB for C1 = 256) instead of
// 32 x (C1 + C2) (99 KB for the head's [256 -> 512 -> 1024] group): TWO workgroups of four waves share a CU, and while one
// of them gathers its first layer, crosses a barrier or pools and stores, the other keeps the matrix pipes busy.  (Phase
// timers of the one-workgroup form on the head's wide group: 104 us per tile of which 68 us are matrix cycles; the rest is
// the first layer's gather, barrier skew and the pooling epilogue, during which a CU with one workgroup idles.)
// Wave w owns columns [32w, 32w + 32) of every second-layer chunk and the quarter [w C3 / 4, (w + 1) C3 / 4) of the third
// layer, whose accumulators (C3 / 128 tiles of 32 x 32 = up to 128 registers) stay live across the chunks: every output
// is still one ascending-k chain (chunks in ascending order), bit-identical to the other forms.
// B fragments: ring of DEPTH register sets of UK k-steps, DEPTH - 1 blocks ahead of their use.
// ------------------------------------------------------------------------------------------------------------------------
template <int K, int TN, int LDX, int UK, int DEPTH>
__device__ __forceinline__ void stream_layer(const float *__restrict__ X, const __amdgpu_buffer_rsrc_t srd, const uint32_t voff,
                                             const int ldw_bytes, const int soff0, f32x16 (&acc)[TN], const int l31, const int kh) {
  constexpr int VW = TN >= 4 ? 4 : TN;
  constexpr int NV = TN / VW;
  constexpr int KS = K / 2;
  constexpr int NB = KS / UK;
  static_assert(KS % UK == 0 && NB % DEPTH == 0 && NB >= DEPTH, "ring structure");
  typedef typename BVec<VW>::T bvec;
  bvec b[DEPTH][UK][NV];
  auto fetch = [&](bvec (&bs)[UK][NV], int blk) {
#pragma unroll
    for (int u = 0; u < UK; ++u)
#pragma unroll
      for (int v = 0; v < NV; ++v) bs[u][v] = load_b<VW>(srd, voff, soff0 + D6_KOFF(2 * (blk * UK + u) * ldw_bytes) + 128 * VW * v);
  };
  const float *xa = X + l31 * LDX + kh;
  float a[UK];
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) fetch(b[d], d);
#pragma unroll
  for (int u = 0; u < UK; ++u) a[u] = xa[2 * u];
#pragma unroll 1
  for (int blk = 0; blk < NB; blk += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int cur = blk + d;
      // unconditional (the tail re-reads the last block): a conditional fetch makes the compiler's s_waitcnt vmcnt
      // accounting assume the shorter queue and drain the ring at every step
      fetch(b[(d + DEPTH - 1) % DEPTH], cur + DEPTH - 1 < NB ? cur + DEPTH - 1 : NB - 1);
      float an[UK];
      const int nxt = cur + 1 < NB ? cur + 1 : cur;      // the A fragment of the next block, ahead of this block's MFMAs
#pragma unroll
      for (int u = 0; u < UK; ++u) an[u] = xa[2 * (nxt * UK + u)];
#pragma unroll
      for (int u = 0; u < UK; ++u)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bget<VW>(b[d][u][j / VW], j % VW), acc[j], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < UK; ++u) a[u] = an[u];
    }
  }
}

template <int C1, int C2, int C3, bool COMPACT>
__global__ __launch_bounds__(256, 2) void mlp_group_stream_kernel(const GroupArgs g) {
  D6_GEMM_PRIO_APPLY();
  constexpr int CH = 128;                          // second-layer columns per chunk = third-layer k per chunk
  constexpr int NCH = C2 / CH;
  constexpr int LD1 = C1 + 1, LDY = CH + 1;
  constexpr int TN3 = C3 / 128;                    // accumulator tiles per wave in the third layer
  static_assert(C2 % CH == 0 && NCH >= 2 && TN3 >= 2, "chunk structure");
  extern __shared__ float lds[];
  float *X1 = lds;
  float *Y0 = lds + 32 * LD1;
  float *Y1 = Y0 + 32 * LDY;
  int *tags = reinterpret_cast<int *>(Y1 + 32 * LDY);             // 2 x 32 row tags, by tile parity
  int it = 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  float *scr = reinterpret_cast<float *>(tags + 64) + wave * (64 * (TN3 >= 4 ? 4 : TN3));   // wave-private: g_store_group
  const int live_tiles = (COMPACT ? g.hdr[0] : g.rows) / 32;
  if ((int)blockIdx.x >= live_tiles) return;
  int h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0;
  if (COMPACT) { h1 = g.hdr[1]; h2 = g.hdr[2]; h3 = g.hdr[3]; h4 = g.hdr[4]; h5 = g.hdr[5]; }

  const __amdgpu_buffer_rsrc_t srd2 = __builtin_amdgcn_make_buffer_rsrc((void *)g.w2, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd3 = __builtin_amdgcn_make_buffer_rsrc((void *)g.w3, 0, 0xffffffff, 0x00020000);
  const uint32_t voff2 = (uint32_t)(kh * g.ldw2 + 32 * wave + l31) * 4u;                          // + 128 c columns per chunk
  const uint32_t voff3 = (uint32_t)(kh * g.ldw3 + wave * (C3 / 4) + tile_col<TN3>(0, l31)) * 4u;   // + 128 c rows per chunk
  float sh3[TN3];
#pragma unroll
  for (int j = 0; j < TN3; ++j) sh3[j] = g.s3[wave * (C3 / 4) + tile_col<TN3>(j, l31)];

  // one chunk of the second layer: Y[:, 32w .. 32w + 31] = relu(X1 W2[:, 128 c + 32 w ..] + s2)
  auto second = [&](const int c, float *__restrict__ Y) {
    const float sh2 = g.s2[CH * c + 32 * wave + l31];
    f32x16 acc2[1];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[0][e] = 0.f;
    stream_layer<C1, 1, LD1, 8, 4>(X1, srd2, voff2, g.ldw2 * 4, CH * c * 4, acc2, l31, kh);
    float *yc = Y + 32 * wave + l31;
#pragma unroll
    for (int e = 0; e < 16; ++e) yc[((e & 3) + 8 * (e >> 2) + 4 * kh) * LDY] = d6_relu(acc2[0][e] + sh2);
  };

  __shared__ int next_tile_s[2];
  int *const ticket = COMPACT ? g.ticket : nullptr;
  for (int tile = blockIdx.x; tile < live_tiles;) {
    int drawn = 0;
    if (ticket && tid == 0) drawn = g_draw_ticket(ticket, (int)gridDim.x);     // tiles by ticket: see g_draw_ticket
    const int par = it & 1;
    int *tagbuf = tags + 32 * (it++ & 1);
    int e0, e1;
    group_row_entry<COMPACT, 8>(g, tile, tid, e0, e1);
    group_layer1<C1, COMPACT, 8>(g, tile, tid, X1, tagbuf, e0, e1);
    if (ticket && tid == 0) next_tile_s[par] = drawn;
    __syncthreads();
    const int next_tile = ticket ? next_tile_s[par] : tile + (int)gridDim.x;
    f32x16 acc[TN3];
#pragma unroll
    for (int j = 0; j < TN3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    second(0, Y0);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      float *Yc = (c & 1) ? Y1 : Y0;
      float *Yn = (c & 1) ? Y0 : Y1;
      // third layer, k = 128 c .. 128 c + 127, for all of this wave's columns
      stream_layer<CH, TN3, LDY, 16 / TN3 >= 1 ? 16 / TN3 : 1, 4>(Yc, srd3, voff3, g.ldw3 * 4, CH * c * g.ldw3 * 4, acc, l31, kh);
      if (c + 1 < NCH) {
        second(c + 1, Yn);            // the other buffer: last read in chunk c - 1, a barrier ago
        __syncthreads();
      }
    }
    group_pool_store<TN3, COMPACT>(g, tile, acc, sh3, wave * (C3 / 4), l31, kh, h1, h2, h3, h4, h5, tagbuf, scr);
    // no barrier here: X1 is rewritten by the next tile's layer 1, every wave is past its last second-layer chunk (the
    // barrier above); Y0 is rewritten after the next tile's first barrier, Y1 two barriers later
    tile = next_tile;
  }
  if (ticket && tid == 0) g_leave(ticket, min((int)gridDim.x, live_tiles));
}

template <int C1, int C2, int C3, bool COMPACT>
int launch_group_stream(const GroupArgs &g, hipStream_t stream) {
  constexpr int kVW3 = (C3 / 128) >= 4 ? 4 : (C3 / 128);
  const size_t lds_bytes = sizeof(float) * (32 * (size_t)(C1 + 1 + 2 * 129) + 64 + 4 * 64 * kVW3);   // + tags + store scratch
  DET6D_MAX_DYNAMIC_LDS((mlp_group_stream_kernel<C1, C2, C3, COMPACT>), lds_bytes);
  int per_cu = (int)((160 * 1024) / lds_bytes);      // 2 for the head's groups (66 KB), 3 for SA3's (52 KB)
  per_cu = per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu);
  int blocks = g.rows / 32;
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
  hipLaunchKernelGGL((mlp_group_stream_kernel<C1, C2, C3, COMPACT>), dim3(blocks), dim3(256), lds_bytes, stream, g);
  return det6d_check_launch("det6d_mlp_group3 (streaming)");
}

template <int C1, int C2, int C3, bool COMPACT, int NW>
int launch_group(const GroupArgs &g, hipStream_t stream) {
  constexpr int kTN3 = C3 / (32 * NW);
  const size_t lds_bytes = sizeof(float) * (32 * (size_t)(C1 + 1 + C2 + 1) + 64 + NW * 64 * (kTN3 >= 4 ? 4 : kTN3));   // + tags + store scratch
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void *)mlp_group_kernel<C1, C2, C3, COMPACT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_set = true;
  }
  int per_cu = (int)((160 * 1024) / lds_bytes);
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  int blocks = g.rows / 32;
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
#ifdef DET6D_EXPERIMENTS
  static bool phase_set = false;
  if (!phase_set) {
    const int pc2 = det6d_env_int("DET6D_GROUP_PHASE_C2", 512), pc3 = det6d_env_int("DET6D_GROUP_PHASE_C3", 1024);
    hipMemcpyToSymbol(HIP_SYMBOL(d6_group_phase_c2), &pc2, sizeof(int));
    hipMemcpyToSymbol(HIP_SYMBOL(d6_group_phase_c3), &pc3, sizeof(int));
    phase_set = true;
  }
  static const int whatif = det6d_env_int("DET6D_GROUP_WHATIF", 0);
  static bool whatif_set = false;
  if (whatif && !whatif_set) {
    const int zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(d6_group_kmul), &zero, sizeof(int));
    whatif_set = true;
  }
#endif
  hipLaunchKernelGGL((mlp_group_kernel<C1, C2, C3, COMPACT, NW>), dim3(blocks), dim3(64 * NW), lds_bytes, stream, g);
  return det6d_check_launch("det6d_mlp_group3");
}

}  // namespace

#ifdef DET6D_EXPERIMENTS
// timing experiments only: reads and clears the phase timers
extern "C" __attribute__((visibility("default"))) int det6d_dbg_group_phase(unsigned long long *out_host) {
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(d6_group_phase), sizeof(unsigned long long) * 26) != hipSuccess) return -1;
  const unsigned long long zero[26] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(d6_group_phase), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

// widths this kernel is built for (Det6D's SA3 and head-SA groups)
static bool group_widths_ok(int c1, int c2, int c3) {
  return (c1 == 128 && c2 == 128 && c3 == 256) || (c1 == 128 && c2 == 256 && c3 == 256) || (c1 == 256 && c2 == 256 && c3 == 512) ||
         (c1 == 256 && c2 == 512 && c3 == 1024);
}

DET6D_API int det6d_mlp_group3_supported(int c1, int c2, int c3, int ns, int compact) {
  return group_widths_ok(c1, c2, c3) && (compact || ns == 16 || ns == 32) ? 1 : 0;
}

DET6D_API int det6d_mlp_group3(int rows, const float *p, int ldp, int pcol0, const float *w1, int ldw1, const float *s1, int c1,
                               const float *w2, int ldw2, const float *s2, int c2, const float *w3, int ldw3, const float *s3,
                               int c3, const float *pts, int ldpts, const float *ctr, int ldctr, const int *idx, int n, int m,
                               int ns, const int *cnt, int *hdr, const int *crow_p, const int *crow_c, float *y, int ldy,
                               int col0, det6d_stream_t stream) {
  D6_GEMM_PRIO_HOST();
  if (rows < 0 || (rows & 31) || !p || !w1 || !w2 || !w3 || !s1 || !s2 || !s3 || !pts || !ctr || !y) return DET6D_EINVAL;
  if (!group_widths_ok(c1, c2, c3)) return DET6D_EINVAL;
  if ((ldp & 3) || (pcol0 & 3) || (ldw1 & 3) || ldp < pcol0 + c1 || ldw1 < c1 || ldw2 < c2 || ldw3 < c3 || ldpts < 3 || ldctr < 3)
    return DET6D_EINVAL;
  if (((uintptr_t)p | (uintptr_t)w1 | (uintptr_t)s1) & 15) return DET6D_EINVAL;
  if ((size_t)c1 * ldw2 * 4 >= 0xfff00000ull || (size_t)c2 * ldw3 * 4 >= 0xfff00000ull) return DET6D_EINVAL;
  const bool compact = hdr != nullptr;
  if (compact ? (!crow_p || !crow_c || (rows & 127)) : (!idx || !cnt || n <= 0 || m <= 0 || (ns != 16 && ns != 32) || rows % (m * ns)))
    return DET6D_EINVAL;
  if (!compact && ns == 16 && (m & 1)) return DET6D_EINVAL;   // two centres of one tile must share the scene
  if (rows == 0) return DET6D_OK;
  GroupArgs g;
  g.rows = rows; g.p = p; g.ldp = ldp; g.pcol0 = pcol0;
  g.w1 = w1; g.ldw1 = ldw1; g.s1 = s1; g.w2 = w2; g.ldw2 = ldw2; g.s2 = s2; g.w3 = w3; g.ldw3 = ldw3; g.s3 = s3;
  g.pts = pts; g.ldpts = ldpts; g.ctr = ctr; g.ldctr = ldctr;
  g.idx = idx; g.n = n; g.m = m; g.ns = ns; g.cnt = cnt;
  g.hdr = hdr; g.crow_p = crow_p; g.crow_c = crow_c;
  // tiles by ticket on compact lists (hdr[10], hdr[11]: zeroed by the list builder and by the kernels themselves); the
  // experiments build keeps the static walk behind DET6D_GROUP_STATIC=1 for A/B runs
  static const int static_tiles = det6d_env_int("DET6D_GROUP_STATIC", 0);
  g.ticket = (hdr && !static_tiles) ? hdr + 10 : nullptr;
  g.y = y; g.ldy = ldy; g.col0 = col0;
  static const int pre_entries = det6d_env_int("DET6D_GROUP_PRE", 1);
  g.pre = pre_entries;
  g.yvec = (!(ldy & 3) && !(col0 & 3) && !((uintptr_t)y & 15)) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  // waves per 32-row tile: 8 for the head's groups (two waves per SIMD from ONE workgroup: the 99 KB of LDS allow only one
  // workgroup per CU), 4 for the SA3 groups (several workgroups per CU); DET6D_GROUP_WAVES (experiments build) overrides
  static const int nw_env = det6d_env_int("DET6D_GROUP_WAVES", 0);
  // DET6D_GROUP_STREAM (bit mask): 1 = the streaming form (two workgroups per CU) for the head's wide group
  // [256 -> 512 -> 1024], 2 = for its narrow group [256 -> 256 -> 512], 3 = both, 0 = the one-pass form everywhere.  Same
  // bits.  Wide group: within +-1 % of the one-pass form in the pipeline once that form fetched its first weight blocks
  // ahead of the producing phase (12.98 vs 12.87 k scenes/s, ray-cast scenes 6.06 vs 6.09 k).  Narrow group (default):
  // its one-pass form holds 143 registers with eight waves, i.e. one workgroup per CU; the streaming form runs it 8-10 %
  // faster with the chip full (196 -> 180 us, ray-cast scenes 446 -> 400 us) and the pipeline gains 0.5-0.7 %.
  // Round 6, at the 80-scene pass size the bench runs since round 5: in the EXPERIMENTS build the streaming form of the wide
  // group and eight waves per tile for SA3's [128 -> 256 -> 256] group looked like +4.3 % / +5.2 % / together +10.3 %
  // (scripts/r06/gpu_t4.sh) — an artefact: that build's one-pass kernels carry phase timers.  The same A/B in the KNOBS build
  // (the shipped kernels with the switches live, scripts/r06/gpu_t6.sh, interleaved, two repeats): round-5 settings 15 034
  // scenes/s; wide group streaming 15 171 (+0.9 %); SA3 with eight waves 14 732 (-2.0 %); SA3's wide group streaming too (mask
  // 7) 15 114; pipelined linear slab 15 138 (+0.7 %).  Default: mask 3, SA3 stays at four waves.
  static const int stream_form = det6d_env_int("DET6D_GROUP_STREAM", 3);
  if ((stream_form & 1) && c1 == 256 && c2 == 512 && c3 == 1024)
    return compact ? launch_group_stream<256, 512, 1024, true>(g, s) : launch_group_stream<256, 512, 1024, false>(g, s);
  if ((stream_form & 2) && c1 == 256 && c2 == 256 && c3 == 512)
    return compact ? launch_group_stream<256, 256, 512, true>(g, s) : launch_group_stream<256, 256, 512, false>(g, s);
  if ((stream_form & 4) && c1 == 128 && c2 == 256 && c3 == 256)      // SA3's wide group (three workgroups per CU)
    return compact ? launch_group_stream<128, 256, 256, true>(g, s) : launch_group_stream<128, 256, 256, false>(g, s);
#define D6_GROUP(A, B, C, NWD)                                                                        \
  if (c1 == A && c2 == B && c3 == C) {                                                                \
    if ((nw_env ? nw_env : NWD) == 8 && B >= 256)                                                     \
      return compact ? launch_group<A, B, C, true, (B >= 256 ? 8 : 4)>(g, s) : launch_group<A, B, C, false, (B >= 256 ? 8 : 4)>(g, s); \
    return compact ? launch_group<A, B, C, true, 4>(g, s) : launch_group<A, B, C, false, 4>(g, s);    \
  }
  static const int sa3_waves = det6d_env_int("DET6D_GROUP_SA3_WAVES", 4);      // knobs build: 8 = eight waves per tile (slower: see below)
  D6_GROUP(128, 128, 256, 4)
  if (sa3_waves != 8) { D6_GROUP(128, 256, 256, 4) }
  D6_GROUP(128, 256, 256, 8)
  D6_GROUP(256, 256, 512, 8)
  D6_GROUP(256, 512, 1024, 8)
#undef D6_GROUP
  return DET6D_EINVAL;
}
