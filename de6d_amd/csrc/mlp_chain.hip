// mlp_chain.hip — the three pointwise layers of a NARROW set-abstraction group (all widths <= 64:
// the first SA layer of Det6D, [4->16->16->32] and [4->32->32->64] on 0.5 M / 1 M rows per batch)
// fused into one launch: gather -> L1 -> L2 -> L3 -> mask -> max-pool, intermediates never leave
// the CU.
//
// Why a separate kernel: at these widths every layer of the generic GEMM (linear.hip) is bound by
// writing and re-reading (rows x 32) fp32 intermediates through HBM (~0.8 GB per batch) and by six
// launches; the arithmetic is only 7.6 GFLOP.
//
// Wave-autonomous design: one wave64 owns a 32-row tile (= nsample rows of one centre, or two centres
// at nsample 16) through all three layers, so there is NO workgroup barrier after the weights are
// staged.  A layer's 32x32 accumulator tiles (v_mfma_f32_32x32x2_f32: channel on the lane, rows in the
// registers) are written to a wave-private LDS tile laid out [channel][row] (row stride 33 floats:
// conflict-free both ways), which is exactly the k-major image the next layer's A fragment
// (A[i = lane&31][k = lane>>5]) reads with one ds_read_b32.  Weights (<= 13 KB for the three layers)
// sit in LDS for the whole kernel.
//
// Arithmetic is identical to three det6d_linear calls: every output is one ascending-k fmaf chain,
// + shift, ReLU; masked max over the nsample rows.
#include "common.h"
#include <stdlib.h>

namespace {

D6_GEMM_PRIO_DECL

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kChainWaves = 4;
constexpr int kPackTiles = 1;    // compact lists: tiles a wave must get before another workgroup is used (1: 9970, 2: 9910, 4: 9780 scenes/s)
constexpr int kTS = 33;       // row stride of the wave-private [channel][row] tiles
constexpr int kMaxK1 = 8;     // input row width (x,y,z,features + pad)
constexpr int kMaxC = 32;     // hidden widths
constexpr int kMaxC3 = 64;    // output width

struct ChainArgs {
  int rows;                   // b * m * ns
  int n, m, ns;
  const float *a; int lda;    // point rows (B, n, lda)
  const int *idx;             // (B, m, ns)
  const float *ctr; int ldctr;
  const int *cnt;             // (B*m)
  const float *w1, *w2, *w3;  // folded weights, row-major, leading dims ldw1.. (multiples of 4)
  int ldw1, ldw2, ldw3;
  const float *s1, *s2, *s3;  // shifts
  int k1, c1, c2, c3;         // true widths: k1 = lda, c1,c2 <= 32, c3 <= 64
  float *y; int ldy; int col0;
  // compact (ragged) rows (csrc/compact.hip): live row count and class regions in hdr, one point row and one
  // centre per compact row; rows = capacity then
  const int *hdr; const int *crow_p; const int *crow_c;
};

// compact rows: class (= pooling width) of the 32-row tile starting at row0; h[c] = end of the region of class 32 >> c
__device__ __forceinline__ int compact_class(int row0, int h1, int h2, int h3, int h4, int h5) {
  return row0 < h1 ? 32 : row0 < h2 ? 16 : row0 < h3 ? 8 : row0 < h4 ? 4 : row0 < h5 ? 2 : 1;
}
// row (inside a 32-row tile) whose centre owns pooled value qq of a lane in half kh, -1: another lane writes it.
// A lane holds rows 8*qq + 4*kh + (0..3); class 4 groups end inside the lane, wider groups after the lane^32 exchange.
__device__ __forceinline__ int compact_out_row(int s, int qq, int kh) {
  if (s < 4) return -1;          // classes 1, 2: stored at once by compact_store_small
  if (s == 4) return 8 * qq + 4 * kh;
  if (kh) return -1;
  if (s == 8) return 8 * qq;
  if (s == 16) return (qq & 1) ? -1 : 8 * qq;
  return qq == 0 ? 0 : -1;
}
// pooled value of one part of a centre: plain store, or (bit 29 of the row tag: the centre's rows are cut into several
// parts, compact.hip) integer atomic max on the non-negative post-ReLU value into the zeroed buffer
__device__ __forceinline__ void compact_store(float *dst, float val, int tag) {
  if (tag & 0x20000000) __hip_atomic_fetch_max(reinterpret_cast<int *>(dst), __builtin_bit_cast(int, val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *dst = val;
}
// classes 1 and 2: every accumulator (pair) of a 32x32 tile is a group of its own: stored at once (no pending slot).
// tags[e] = centre tag of the row accumulator e of this lane holds (compact_row_tags).
__device__ __forceinline__ void compact_store_small(const ChainArgs &g, const f32x16 &o, int s, const int (&tags)[16], int col, float sh) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    if (s == 2 && (e & 1)) continue;
    const float raw = s == 2 ? d6_vmax(o[e], o[e + 1 < 16 ? e + 1 : e]) : o[e];
    const int tag = tags[e];
    if (tag >= 0) compact_store(g.y + (size_t)(tag & 0x1fffffff) * g.ldy + g.col0 + col, (tag & 0x40000000) ? 0.f : d6_relu(raw + sh), tag);
  }
}
// Row tags of a tile from the lanes that loaded them (lane r and r + 32 hold the tag of row r): accumulator e of a lane in
// half kh is row (e & 3) + 8 (e >> 2) + 4 kh.  A cross-lane read (ds_bpermute, no memory) instead of a global load per
// accumulator: round 2 re-read crow_c inside the epilogue, 16 dependent L2 round trips per column tile, and — the class
// regions being contiguous — the workgroups that owned the tail of the list (classes 2 and 1) ran several times longer
// than the others (the kernels' matrix pipes were busy 33-42 %: profiles/r03_beam_*).
__device__ __forceinline__ void compact_row_tags(int my_tag, int kh, int (&tags)[16]) {
#pragma unroll
  for (int e = 0; e < 16; ++e) tags[e] = __shfl(my_tag, (e & 3) + 8 * (e >> 2) + 4 * kh);
}
// the four 4-row maxima of a lane -> pooled values of class s (in place; v[qq] valid where compact_out_row >= 0)
__device__ __forceinline__ void compact_pool(float (&v)[4], int s) {
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

__device__ __forceinline__ float relu1(float v) { return v > 0.f ? v : 0.f; }

// column of a gathered row [x - cx, y - cy, z - cz, f_0 ..] at position j of the oracle's fma chain (chain_k in
// oracle/det6d_oracle.c): the feature columns first, the three relative coordinates last
__host__ __device__ __forceinline__ int chain_col(int j, int k1) { return j + 3 < k1 ? j + 3 : j + 3 - k1; }

__global__ __launch_bounds__(64 * kChainWaves) void mlp_chain_kernel(const ChainArgs g) {
  D6_GEMM_PRIO_APPLY();
  __shared__ float W1[kMaxK1 * kMaxC];
  __shared__ float W2[kMaxC * kMaxC];
  __shared__ float W3[kMaxC * kMaxC3];
  __shared__ float S1[kMaxC], S2[kMaxC], S3[kMaxC3];
  __shared__ float T[kChainWaves][2][kMaxC * kTS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;

  // ---- stage weights (zero padded to the tile widths) ----
  // layer 1 takes the gathered row in the oracle's chain order (chain_k): feature columns first, relative coordinates last
  for (int t = tid; t < kMaxK1 * kMaxC; t += blockDim.x) {
    const int k = t / kMaxC, c = t % kMaxC;
    W1[t] = (k < g.k1 && c < g.c1) ? g.w1[(size_t)chain_col(k, g.k1) * g.ldw1 + c] : 0.f;
  }
  for (int t = tid; t < kMaxC * kMaxC; t += blockDim.x) {
    const int k = t / kMaxC, c = t % kMaxC;
    W2[t] = (k < g.c1 && c < g.c2) ? g.w2[(size_t)k * g.ldw2 + c] : 0.f;
  }
  for (int t = tid; t < kMaxC * kMaxC3; t += blockDim.x) {
    const int k = t / kMaxC3, c = t % kMaxC3;
    W3[t] = (k < g.c2 && c < g.c3) ? g.w3[(size_t)k * g.ldw3 + c] : 0.f;
  }
  if (tid < kMaxC) { S1[tid] = tid < g.c1 ? g.s1[tid] : 0.f; S2[tid] = tid < g.c2 ? g.s2[tid] : 0.f; }
  if (tid < kMaxC3) S3[tid] = tid < g.c3 ? g.s3[tid] : 0.f;
  __syncthreads();

  float *T1 = T[wave][0], *T2 = T[wave][1];
  const int ntiles = g.rows / 32;
  const int k1e = (g.k1 + 1) & ~1, c1e = (g.c1 + 1) & ~1, c2e = (g.c2 + 1) & ~1;
  const int nt3 = (g.c3 + 31) / 32;

  for (int tile = blockIdx.x * kChainWaves + wave; tile < ntiles; tile += gridDim.x * kChainWaves) {
    const int r = tile * 32 + l31;                 // my row (both lane halves hold the same row)
    const int cj = r / g.ns;
    const int bi = cj / g.m;
    const int p = g.idx[r];
    const float *src = g.a + ((size_t)bi * g.n + p) * g.lda;
    const float *c = g.ctr + (size_t)cj * g.ldctr;
    // ---- layer 1: A' = [xyz - centre, features], K = k1 ----
    float av[kMaxK1];
#pragma unroll
    for (int k = 0; k < kMaxK1; ++k) av[k] = 0.f;
    {
      const float4 v0 = *reinterpret_cast<const float4 *>(src);
      av[0] = v0.x - c[0]; av[1] = v0.y - c[1]; av[2] = v0.z - c[2]; av[3] = v0.w;
      if (g.k1 > 4) {
        const float4 v1 = *reinterpret_cast<const float4 *>(src + 4);
        av[4] = v1.x; av[5] = v1.y; av[6] = v1.z; av[7] = v1.w;
      }
    }
    {
      float t8[kMaxK1];
#pragma unroll
      for (int k = 0; k < kMaxK1; ++k) t8[k] = av[k];
      if (g.k1 == 4) { av[0] = t8[3]; av[1] = t8[0]; av[2] = t8[1]; av[3] = t8[2]; }
      else { av[0] = t8[3]; av[1] = t8[4]; av[2] = t8[5]; av[3] = t8[6]; av[4] = t8[7]; av[5] = t8[0]; av[6] = t8[1]; av[7] = t8[2]; }
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < kMaxK1 / 2; ++s) {
      if (2 * s < k1e) {
        const float a = kh ? av[2 * s + 1] : av[2 * s];
        const float b = W1[(2 * s + kh) * kMaxC + l31];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
    {
      const float sh = S1[l31];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * kh;
        T1[l31 * kTS + row] = relu1(acc[e] + sh);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // ---- layer 2: K = c1 ----
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int s = 0; s < c1e / 2; ++s) {
      const float a = T1[(2 * s + kh) * kTS + l31];
      const float b = W2[(2 * s + kh) * kMaxC + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    {
      const float sh = S2[l31];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * kh;
        T2[l31 * kTS + row] = relu1(acc[e] + sh);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // ---- layer 3: K = c2, N = c3 (1 or 2 column tiles), then mask + max over the nsample rows ----
    f32x16 acc3[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc3[j][e] = 0.f;
    for (int s = 0; s < c2e / 2; ++s) {
      const float a = T2[(2 * s + kh) * kTS + l31];
      const float b0 = W3[(2 * s + kh) * kMaxC3 + l31];
      acc3[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc3[0], 0, 0, 0);
      if (nt3 > 1) {
        const float b1 = W3[(2 * s + kh) * kMaxC3 + 32 + l31];
        acc3[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc3[1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j >= nt3) break;
      const int col = 32 * j + l31;
      const bool cok = col < g.c3;
      const float sh = S3[col];
      float q[4];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        float mx = relu1(acc3[j][4 * qq] + sh);
#pragma unroll
        for (int e = 1; e < 4; ++e) {
          const float v = relu1(acc3[j][4 * qq + e] + sh);
          mx = v > mx ? v : mx;
        }
        const float o = __shfl_xor(mx, 32);
        q[qq] = o > mx ? o : mx;
      }
      if (g.ns == 32) {
        float mx = q[0];
        mx = q[1] > mx ? q[1] : mx; mx = q[2] > mx ? q[2] : mx; mx = q[3] > mx ? q[3] : mx;
        if (cok && kh == 0) g.y[(size_t)tile * g.ldy + g.col0 + col] = (g.cnt[tile] > 0) ? mx : 0.f;
      } else {  // ns == 16: two centres per tile
        const float m0 = q[1] > q[0] ? q[1] : q[0];
        const float m1 = q[3] > q[2] ? q[3] : q[2];
        if (cok && kh == 0) {
          g.y[(size_t)(2 * tile) * g.ldy + g.col0 + col] = (g.cnt[2 * tile] > 0) ? m0 : 0.f;
          g.y[(size_t)(2 * tile + 1) * g.ldy + g.col0 + col] = (g.cnt[2 * tile + 1] > 0) ? m1 : 0.f;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();   // T1/T2 are rewritten by the next tile
  }
}

// ---------------------------------------------------------------------------------------------------
// Register-resident variant for the widths Det6D's first SA layer uses ([4->16->16->32], [4->32->32->64]).
//
// Layers 1 and 2 are computed TRANSPOSED, D = W^T X^T: the weight fragment is the MFMA A operand (channel on
// lane&31), the activations the B operand (row on lane&31).  The accumulator then holds, per lane, 16
// CHANNELS of ONE row — and after ReLU that is, up to a lane-half exchange, exactly the B (or A) operand
// fragment of the next layer: `v_permlane32_swap` on register pairs turns {half0: ch 8G+0..3, half1: ch
// 8G+4..7} into per-step fragments {half0: ch 2s, half1: ch 2s+1}.  No LDS round trip, no barrier, no
// transposing stores; all weight fragments of the three layers stay in VGPRs (52 for the wide group).
// The folded-BN shift of layers 1-2 rides on one extra MFMA step (a = shift on the k0 half, b = 1):
// fma(shift, 1, acc) == acc + shift exactly, added last like the oracle does.  Layer 3 runs in the normal
// orientation (same fragments, operands swapped back) so that the max over the nsample rows is the cheap
// in-register / one-shuffle epilogue and the store is coalesced.
// Every output is still ONE ascending-k fma chain: bit-identical to det6d_linear x 3.
template <int C1, int C2, int C3, int NS, bool COMPACT = false>
__global__ __launch_bounds__(256) void mlp_chain_reg_kernel(const ChainArgs g) {
  D6_GEMM_PRIO_APPLY();
  constexpr int S1 = 2;            // k1 = 4: [dx, dy, dz, f]
  constexpr int S2 = C1 / 2, S3 = C2 / 2, NT3 = C3 / 32;
  const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
  int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int n_waves = (gridDim.x * blockDim.x) >> 6;
  int pk_end = 0;
  if (COMPACT) {
    // the live tiles of a compact list go round the workgroups that are needed (one tile per wave at least): tile t -> wave
    // t mod (active waves).  The class regions are contiguous (32-row parts first, single rows last) and the small classes
    // have the dearest epilogue, so contiguous chunks per workgroup (round 2) left the tail of the list to a few workgroups.
    const int wpw = blockDim.x >> 6, live = g.hdr[0] / 32;
    int nb = (live + wpw * kPackTiles - 1) / (wpw * kPackTiles);
    if (nb > (int)gridDim.x) nb = gridDim.x;
    if ((int)blockIdx.x >= nb) return;
    n_waves = nb * wpw;
    pk_end = live;
  }

  // weight fragments: lane (channel / column = l31, k = 2s + kh); one more step per transposed layer for the shift
  float wf1[S1 + 1], wf2[S2 + 1], wf3[NT3][S3];
#pragma unroll
  for (int s = 0; s < S1; ++s) wf1[s] = l31 < C1 ? g.w1[(size_t)chain_col(2 * s + kh, 4) * g.ldw1 + l31] : 0.f;   // rows 3,0 | 1,2
  wf1[S1] = (kh == 0 && l31 < C1) ? g.s1[l31] : 0.f;
#pragma unroll
  for (int s = 0; s < S2; ++s) wf2[s] = l31 < C2 ? g.w2[(size_t)(2 * s + kh) * g.ldw2 + l31] : 0.f;
  wf2[S2] = (kh == 0 && l31 < C2) ? g.s2[l31] : 0.f;
#pragma unroll
  for (int j = 0; j < NT3; ++j)
#pragma unroll
    for (int s = 0; s < S3; ++s) wf3[j][s] = g.w3[(size_t)(2 * s + kh) * g.ldw3 + 32 * j + l31];
  float sh3[NT3];
#pragma unroll
  for (int j = 0; j < NT3; ++j) sh3[j] = g.s3[32 * j + l31];
  const float one_k0 = kh == 0 ? 1.f : 0.f;

  int h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0;
  if (COMPACT) { h1 = g.hdr[1]; h2 = g.hdr[2]; h3 = g.hdr[3]; h4 = g.hdr[4]; h5 = g.hdr[5]; }
  const int ntiles = COMPACT ? pk_end : g.rows / 32;   // end of this wave's tile range
  const int *nb_idx = COMPACT ? g.crow_p : g.idx;   // per-row neighbour: point index inside the scene / global point row
  // the tile index is wave-uniform: kept in SGPRs so that the batch index (a division by m) is scalar work
  int tile = __builtin_amdgcn_readfirstlane(wave_global);
  if (tile >= ntiles) return;
  // software pipeline: the neighbour index of the tile after next and the point row of the next tile are in
  // flight while this tile computes (index -> row is a dependent pair of loads)
  struct TileIn { float4 row; float cx, cy, cz; int cnt0, cnt1; int oc[4]; int tag; };
  // list entries of a tile: neighbour index, and on compact lists the centre tag of the row and the tags of the pooled
  // rows.  They are the FIRST of the gather's two dependent loads and travel a whole tile ahead of the rows / centres they
  // address (round 2 fetched the tags together with the rows: every iteration then waited for an L2 round trip between
  // the tag load and the centre load — the matrix pipe of this kernel was busy 39 % of the time on ray-cast scenes)
  struct TileIdx { int p, cj; };
  auto fetch_idx = [&](int t) {
    TileIdx ix;
    ix.p = nb_idx[t * 32 + l31];
    ix.cj = COMPACT ? g.crow_c[t * 32 + l31] : 0;
    return ix;
  };
  auto fetch = [&](int t, const TileIdx &ix) {   // t wave-uniform; for NS == 16 the two centres of a tile share the batch (m even)
    TileIn in;
    const int p = ix.p;
    if (COMPACT) {
      in.row = *reinterpret_cast<const float4 *>(g.a + (size_t)p * 4);
      const int cj = ix.cj;
      const float *c = g.ctr + (size_t)(cj < 0 ? 0 : cj & 0x1fffffff) * g.ldctr;
      in.cx = c[0]; in.cy = c[1]; in.cz = c[2];
      in.cnt0 = in.cnt1 = 0;
      in.tag = cj;
      // tags of the rows whose centres own this lane's pooled values: held by lane `row` (no second load of the list)
      const int sc = compact_class(t * 32, h1, h2, h3, h4, h5);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int r = compact_out_row(sc, qq, kh);
        const int tg = __shfl(cj, r < 0 ? 0 : r);
        in.oc[qq] = r >= 0 ? tg : -1;
      }
      return in;
    }
    in.tag = 0;
    const int c0 = NS == 32 ? t : 2 * t;             // first centre of the tile (scalar)
    const int bi = c0 / g.m;                         // scalar division
    const int cj = NS == 32 ? c0 : c0 + (l31 >> 4);
    in.row = *reinterpret_cast<const float4 *>(g.a + ((size_t)bi * g.n + p) * 4);
    const float *c = g.ctr + (size_t)cj * g.ldctr;
    in.cx = c[0]; in.cy = c[1]; in.cz = c[2];
    in.cnt0 = g.cnt[c0];
    in.cnt1 = NS == 32 ? 0 : g.cnt[c0 + 1];
    return in;
  };
  TileIn nxt = fetch(tile, fetch_idx(tile));
  TileIdx ix_next = tile + n_waves < ntiles ? fetch_idx(tile + n_waves) : TileIdx{0, 0};
  // results are stored one iteration late, BEFORE the next prefetch is issued: the wait for the prefetched
  // inputs at the top of an iteration then never waits for this tile's stores (vmcnt counts in order)
  float pend[NT3][COMPACT ? 4 : 2];
  int pend_oc[4] = {-1, -1, -1, -1};
  int pend_tile = -1;
  auto flush = [&]() {
#pragma unroll
    for (int j = 0; j < NT3; ++j) {
      const int col = 32 * j + l31;
      if (COMPACT) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          if (pend_oc[qq] >= 0) compact_store(g.y + (size_t)(pend_oc[qq] & 0x1fffffff) * g.ldy + g.col0 + col, pend[j][qq], pend_oc[qq]);
      } else if (NS == 32) {
        g.y[(size_t)pend_tile * g.ldy + g.col0 + col] = pend[j][0];
      } else {
        g.y[(size_t)(2 * pend_tile) * g.ldy + g.col0 + col] = pend[j][0];
        g.y[(size_t)(2 * pend_tile + 1) * g.ldy + g.col0 + col] = pend[j][1];
      }
    }
  };
  for (; tile < ntiles; tile += n_waves) {
    const TileIn cur = nxt;
    if (pend_tile >= 0) flush();
    if (tile + n_waves < ntiles) {
      nxt = fetch(tile + n_waves, ix_next);
      if (tile + 2 * n_waves < ntiles) ix_next = fetch_idx(tile + 2 * n_waves);
    }
    const float4 v0 = cur.row;
    const float cx = cur.cx, cy = cur.cy, cz = cur.cz;
    const float x0 = v0.x - cx, x1 = v0.y - cy, x2 = v0.z - cz, x3 = v0.w;

    // ---- layer 1 (transposed): acc[e] = channel (e&3) + 8*(e>>2) + 4*kh of row l31 ----
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf1[0], kh ? x0 : x3, acc, 0, 0, 0);   // chain order: f, dx, dy, dz
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf1[1], kh ? x2 : x1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf1[2], one_k0, acc, 0, 0, 0);
    float frag[16];   // frag[s] = activation fragment of k-step s: {half0: channel 2s, half1: channel 2s + 1}
    auto to_fragments = [&](const f32x16 &a, int nreg) {
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        if (4 * G >= nreg) break;
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = d6_relu(a[4 * G + e]);
        // inline asm: this compiler drops the SECOND result of __builtin_amdgcn_permlane32_swap (seen in the ISA:
        // the source register is reused right after the swap).  a = [a.lo | b.lo], b = [a.hi | b.hi] afterwards.
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                     : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
        frag[4 * G + 0] = t[0];   // channels 8G + 0, 1
        frag[4 * G + 1] = t[2];   // channels 8G + 2, 3
        frag[4 * G + 2] = t[1];   // channels 8G + 4, 5
        frag[4 * G + 3] = t[3];   // channels 8G + 6, 7
      }
    };
    to_fragments(acc, S2);
    // ---- layer 2 (transposed) ----
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < S2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf2[s], frag[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf2[S2], one_k0, acc, 0, 0, 0);
    to_fragments(acc, S3);
    // ---- layer 3 (rows in the registers again), shift + ReLU + mask + max over the nsample rows ----
#pragma unroll
    for (int j = 0; j < NT3; ++j) {
      f32x16 o;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
      for (int s = 0; s < S3; ++s) o = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[s], wf3[j][s], o, 0, 0, 0);
      const int col = 32 * j + l31;
      // max over the rows on the raw accumulators, shift + ReLU on the pooled value (monotone: same result)
      float q[4];
      if (COMPACT) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) q[qq] = d6_vmax(d6_vmax(o[4 * qq], o[4 * qq + 1]), d6_vmax(o[4 * qq + 2], o[4 * qq + 3]));
        const int sc = compact_class(tile * 32, h1, h2, h3, h4, h5);
        if (sc < 4) {
          int tags[16];
          compact_row_tags(cur.tag, kh, tags);
          compact_store_small(g, o, sc, tags, col, sh3[j]);
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) pend[j][qq] = 0.f;
          continue;
        }
        compact_pool(q, sc);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) pend[j][qq] = (cur.oc[qq] & 0x40000000) ? 0.f : d6_relu(q[qq] + sh3[j]);
        continue;
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float mq = d6_vmax(d6_vmax(o[4 * qq], o[4 * qq + 1]), d6_vmax(o[4 * qq + 2], o[4 * qq + 3]));
        q[qq] = d6_vmax(mq, __shfl_xor(mq, 32));
      }
      if (NS == 32) {
        const float mx = d6_relu(d6_vmax(d6_vmax(q[0], q[1]), d6_vmax(q[2], q[3])) + sh3[j]);
        pend[j][0] = (cur.cnt0 > 0) ? mx : 0.f;   // both lane halves hold the pooled value: all 64 lanes store it
      } else {
        const float m0 = d6_relu(d6_vmax(q[0], q[1]) + sh3[j]);
        const float m1 = d6_relu(d6_vmax(q[2], q[3]) + sh3[j]);
        pend[j][0] = (cur.cnt0 > 0) ? m0 : 0.f;
        pend[j][1] = (cur.cnt1 > 0) ? m1 : 0.f;
      }
    }
    pend_tile = tile;
    if (COMPACT) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) pend_oc[qq] = cur.oc[qq];
    }
  }
  flush();
}

// N k-steps of one 32 x 32 accumulator tile whose WEIGHT fragments come from LDS (w[u * stride], stride a compile-time
// multiple of the row length) and whose activation fragments x[0..N) sit in registers.  The fragments of block b + 1 (8
// k-steps) are requested BEFORE the MFMAs of block b and land in registers of their own: left to itself the compiler reads
// every fragment (pair) into the same register right before its use and waits for the LDS round trip with one MFMA in
// flight — the matrix pipe of mlp_chain_wide_kernel was busy 33-42 % of the time (rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES,
// profiles/r03_beam_*).  W_IS_A: the weights are the A operand (transposed layers) or the B operand (layer 3).
template <int N, int STRIDE, bool W_IS_A>
__device__ __forceinline__ void mfma_steps_lds(const float *__restrict__ w, const float *x, f32x16 &acc) {
  constexpr int UB = 8, NB = (N + UB - 1) / UB;
  float wb[2][UB];
#pragma unroll
  for (int u = 0; u < UB; ++u)
    if (u < N) wb[0][u] = w[u * STRIDE];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b + 1 < NB) {
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if ((b + 1) * UB + u < N) wb[(b + 1) & 1][u] = w[((b + 1) * UB + u) * STRIDE];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (b * UB + u < N) {
        if (W_IS_A) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[b & 1][u], x[b * UB + u], acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[b * UB + u], wb[b & 1][u], acc, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------------------------------
// Wide register chain: the same transposed-accumulator scheme for the second SA layer's groups
// ([68->64->64->128], [68->64->96->128]): 264 / 361 MFMAs per 32-row tile.  The weights (66-92 KB with the
// shift appended as one more k-row pair) sit in LDS, row-major [k][C], which is conflict-free for the A
// fragment of the transposed layers (channel on lane&31) and for the B fragment of layer 3 alike; they
// are read one ds_read_b32 per MFMA (LDS instructions do not take vector-ALU slots).  The gathered input
// row never touches LDS either: lane (row, kh) loads x[row][2s + kh] directly, 34 dword loads with
// immediate offsets, issued for the NEXT tile as soon as layer 1 of the current one has consumed them.
// Per tile ~250 vector-ALU ops (ReLU, swaps, pooling) against 17-23 k cycles of matrix work.
template <int C2, int NS, bool COMPACT = false>
__global__ __launch_bounds__(512) void mlp_chain_wide_kernel(const ChainArgs g) {
  D6_GEMM_PRIO_APPLY();
  constexpr int K1 = 68, C1 = 64, C3 = 128;
  constexpr int S1 = K1 / 2, S2 = C1 / 2, S3 = C2 / 2;
  constexpr int T1 = C1 / 32, T2 = C2 / 32, T3 = C3 / 32;
  extern __shared__ float lds[];
  float *W1 = lds;                          // (K1 + 2) x C1 : rows K1, K1+1 = shift, 0
  float *W2 = W1 + (K1 + 2) * C1;           // (C1 + 2) x C2
  float *W3 = W2 + (C1 + 2) * C2;           // C2 x C3
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
  // compact lists: the grid is sized for the capacity but the list holds a fraction of it: only the workgroups that get at
  // least kPackTiles tiles per wave stay (the others leave before staging the 66-92 KB of weights); the live tiles are
  // dealt round them, tile t -> wave t mod (active waves).
  int pk_stride = 0, pk_first = 0, pk_end = 0;
  if (COMPACT) {   // tile t -> wave t mod (active waves): every workgroup sees the same mix of classes (see mlp_chain_reg_kernel)
    const int wpw = blockDim.x >> 6, live = g.hdr[0] / 32;
    int nb = (live + wpw * kPackTiles - 1) / (wpw * kPackTiles);
    if (nb > (int)gridDim.x) nb = gridDim.x;
    if ((int)blockIdx.x >= nb) return;
    pk_stride = nb * wpw;
    pk_first = blockIdx.x * wpw + (tid >> 6);
    pk_end = live;
  }
  // staging with 16-byte loads (all leading dimensions and widths are multiples of 4)
  auto stage = [&](float *dst, const float *w, int ldw, const float *shift, int k_rows, int cols, bool chain_order) {
    const int c4 = cols / 4;
    for (int t = tid; t < (k_rows + (shift ? 2 : 0)) * c4; t += blockDim.x) {
      const int k = t / c4, c = 4 * (t % c4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < k_rows) v = *reinterpret_cast<const float4 *>(w + (size_t)(chain_order ? chain_col(k, k_rows) : k) * ldw + c);
      else if (k == k_rows) v = *reinterpret_cast<const float4 *>(shift + c);
      *reinterpret_cast<float4 *>(dst + k * cols + c) = v;
    }
  };
  stage(W1, g.w1, g.ldw1, g.s1, K1, C1, true);    // layer 1 in the oracle's chain order: features, then dx, dy, dz
  stage(W2, g.w2, g.ldw2, g.s2, C1, C2, false);
  stage(W3, g.w3, g.ldw3, nullptr, C2, C3, false);
  float sh3[T3];
#pragma unroll
  for (int j = 0; j < T3; ++j) sh3[j] = g.s3[32 * j + l31];
  __syncthreads();

  const float one_k0 = kh == 0 ? 1.f : 0.f;
  const int wave_global = COMPACT ? pk_first : (int)((blockIdx.x * blockDim.x + tid) >> 6);
  const int n_waves = COMPACT ? pk_stride : (int)((gridDim.x * blockDim.x) >> 6);   // tile stride of this wave
  int h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0;
  if (COMPACT) { h1 = g.hdr[1]; h2 = g.hdr[2]; h3 = g.hdr[3]; h4 = g.hdr[4]; h5 = g.hdr[5]; }
  const int ntiles = COMPACT ? pk_end : g.rows / 32;                                 // end of this wave's tile range
  int tile = __builtin_amdgcn_readfirstlane(wave_global);
  if (tile >= ntiles) return;

  float xin[S1 + 1];        // [S1] = the shift step's activation fragment (1 on the k0 half)
  float csub0, csub1;
  int cnt0, cnt1;
  int oc_n[4] = {-1, -1, -1, -1};
  // list entries of a tile (point row, centre tag, tags of the pooled rows): the FIRST of the gather's two dependent
  // loads, requested a whole tile ahead of the rows they address (fetch_entries(t + 2 strides) while tile t computes)
  int e_p = 0, e_c = 0, tag_n = 0;
  auto fetch_entries = [&](int t) {
    if (COMPACT) {
      e_p = g.crow_p[t * 32 + l31];
      e_c = g.crow_c[t * 32 + l31];
    } else {
      e_p = g.idx[t * 32 + l31];
    }
  };
  auto fetch = [&](int t) {   // t wave-uniform; consumes the entries fetch_entries(t) left in e_p / e_c
    if (COMPACT) {
      const int p = e_p;
      const float *row = g.a + (size_t)p * K1;
      const float *src = row + 3 + kh;          // chain position j = 2s + kh reads column j + 3 (features) ...
#pragma unroll
      for (int s = 0; s < S1 - 2; ++s) xin[s] = src[2 * s];
      xin[S1 - 2] = row[kh ? 0 : K1 - 1];       // ... then (pad, x) and (y, z): chain_col(64..67, 68) = 67, 0, 1, 2
      xin[S1 - 1] = row[kh ? 2 : 1];
      const int cj = e_c;
      const float *c = g.ctr + (size_t)(cj < 0 ? 0 : cj & 0x1fffffff) * g.ldctr;
      csub0 = kh ? c[0] : 0.f;
      csub1 = kh ? c[2] : c[1];
      cnt0 = cnt1 = 0;
      tag_n = cj;
      const int sc = compact_class(t * 32, h1, h2, h3, h4, h5);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {   // tags of the pooled rows from the lanes that hold them
        const int r = compact_out_row(sc, qq, kh);
        const int tg = __shfl(cj, r < 0 ? 0 : r);
        oc_n[qq] = r >= 0 ? tg : -1;
      }
      return;
    }
    const int c0 = NS == 32 ? t : 2 * t;
    const int bi = c0 / g.m;
    const int cj = NS == 32 ? c0 : c0 + (l31 >> 4);
    const int p = e_p;
    const float *row = g.a + ((size_t)bi * g.n + p) * K1;
    const float *src = row + 3 + kh;
#pragma unroll
    for (int s = 0; s < S1 - 2; ++s) xin[s] = src[2 * s];
    xin[S1 - 2] = row[kh ? 0 : K1 - 1];
    xin[S1 - 1] = row[kh ? 2 : 1];
    const float *c = g.ctr + (size_t)cj * g.ldctr;
    csub0 = kh ? c[0] : 0.f;          // chain positions 64 | 65: pad | x
    csub1 = kh ? c[2] : c[1];         // chain positions 66 | 67: y | z
    cnt0 = g.cnt[c0];
    cnt1 = NS == 32 ? 0 : g.cnt[c0 + 1];
  };
  auto to_fragments = [&](const f32x16 &a, float *frag) {   // 16 accumulator registers -> 16 k-step fragments
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      float t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = d6_relu(a[4 * G + e]);
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                   : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
      frag[4 * G + 0] = t[0]; frag[4 * G + 1] = t[2]; frag[4 * G + 2] = t[1]; frag[4 * G + 3] = t[3];
    }
  };

  float pend[T3][COMPACT ? 4 : 2];
  int pend_oc[4] = {-1, -1, -1, -1};
  int pend_tile = -1;
  auto flush = [&]() {
#pragma unroll
    for (int j = 0; j < T3; ++j) {
      const int col = 32 * j + l31;
      if (COMPACT) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          if (pend_oc[qq] >= 0) compact_store(g.y + (size_t)(pend_oc[qq] & 0x1fffffff) * g.ldy + g.col0 + col, pend[j][qq], pend_oc[qq]);
      } else if (NS == 32) {
        g.y[(size_t)pend_tile * g.ldy + g.col0 + col] = pend[j][0];
      } else {
        g.y[(size_t)(2 * pend_tile) * g.ldy + g.col0 + col] = pend[j][0];
        g.y[(size_t)(2 * pend_tile + 1) * g.ldy + g.col0 + col] = pend[j][1];
      }
    }
  };

  fetch_entries(tile);
  fetch(tile);
  if (tile + n_waves < ntiles) fetch_entries(tile + n_waves);
  for (; tile < ntiles; tile += n_waves) {
    if (pend_tile >= 0) flush();
    const int my_cnt0 = cnt0, my_cnt1 = cnt1, my_tag = tag_n;
    int my_oc[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) my_oc[qq] = oc_n[qq];
    xin[S1 - 2] = xin[S1 - 2] - csub0;     // pad | dx
    xin[S1 - 1] = xin[S1 - 1] - csub1;     // dy | dz
    xin[S1] = one_k0;
    // ---- layer 1 (transposed): K1 -> C1, the shift as one more k-step (rows K1, K1 + 1 of the staged matrix) ----
    float f1[S2 + 1];
#pragma unroll
    for (int t = 0; t < T1; ++t) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      mfma_steps_lds<S1 + 1, 2 * C1, true>(W1 + kh * C1 + 32 * t + l31, xin, acc);
      to_fragments(acc, f1 + 16 * t);
    }
    f1[S2] = one_k0;
    // the input registers are free: start the next tile's gather now (its list entries arrived a tile ago), it lands
    // during layers 2 and 3; then request the list entries of the tile after next
    if (tile + n_waves < ntiles) {
      fetch(tile + n_waves);
      if (tile + 2 * n_waves < ntiles) fetch_entries(tile + 2 * n_waves);
    }
    // ---- layer 2 (transposed): C1 -> C2 ----
    float f2[S3];
#pragma unroll
    for (int t = 0; t < T2; ++t) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      mfma_steps_lds<S2 + 1, 2 * C2, true>(W2 + kh * C2 + 32 * t + l31, f1, acc);
      to_fragments(acc, f2 + 16 * t);
    }
    // ---- layer 3 (rows in the registers): C2 -> C3, pool, shift, ReLU, mask ----
#pragma unroll
    for (int j = 0; j < T3; ++j) {
      f32x16 o;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = 0.f;
      mfma_steps_lds<S3, 2 * C3, false>(W3 + kh * C3 + 32 * j + l31, f2, o);
      float q[4];
      if (COMPACT) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) q[qq] = d6_vmax(d6_vmax(o[4 * qq], o[4 * qq + 1]), d6_vmax(o[4 * qq + 2], o[4 * qq + 3]));
        const int sc = compact_class(tile * 32, h1, h2, h3, h4, h5);
        if (sc < 4) {
          int tags[16];
          compact_row_tags(my_tag, kh, tags);
          compact_store_small(g, o, sc, tags, 32 * j + l31, sh3[j]);
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) pend[j][qq] = 0.f;
          continue;
        }
        compact_pool(q, sc);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) pend[j][qq] = (my_oc[qq] & 0x40000000) ? 0.f : d6_relu(q[qq] + sh3[j]);
        continue;
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float mq = d6_vmax(d6_vmax(o[4 * qq], o[4 * qq + 1]), d6_vmax(o[4 * qq + 2], o[4 * qq + 3]));
        q[qq] = d6_vmax(mq, __shfl_xor(mq, 32));
      }
      if (NS == 32) {
        const float mx = d6_relu(d6_vmax(d6_vmax(q[0], q[1]), d6_vmax(q[2], q[3])) + sh3[j]);
        pend[j][0] = my_cnt0 > 0 ? mx : 0.f;
      } else {
        pend[j][0] = my_cnt0 > 0 ? d6_relu(d6_vmax(q[0], q[1]) + sh3[j]) : 0.f;
        pend[j][1] = my_cnt1 > 0 ? d6_relu(d6_vmax(q[2], q[3]) + sh3[j]) : 0.f;
      }
    }
    pend_tile = tile;
    if (COMPACT) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) pend_oc[qq] = my_oc[qq];
    }
  }
  flush();
}

}  // namespace

DET6D_API int det6d_mlp_chain3(int rows, int n, int m, int ns, const float *a, int lda, const int *idx,
                               const float *ctr, int ldctr, const int *cnt, const float *w1, int ldw1,
                               const float *s1, int c1, const float *w2, int ldw2, const float *s2, int c2,
                               const float *w3, int ldw3, const float *s3, int c3, float *y, int ldy, int col0,
                               det6d_stream_t stream) {
  D6_GEMM_PRIO_HOST();
  if (rows < 0 || n <= 0 || m <= 0 || (ns != 16 && ns != 32) || !a || !idx || !ctr || !cnt || !w1 || !w2 || !w3 ||
      !s1 || !s2 || !s3 || !y)
    return DET6D_EINVAL;
  const bool wide = lda == 68 && c1 == 64 && (c2 == 64 || c2 == 96) && c3 == 128 && (ns == 32 || !(m & 1)) && !(ldw1 & 3) &&
                    !(ldw2 & 3) && !(ldw3 & 3) &&
                    !(((uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)w3 | (uintptr_t)s1 | (uintptr_t)s2) & 15) &&
                    !det6d_env_set("DET6D_CHAIN_NO_WIDE");
  if (!wide) {
    if (lda < 4 || lda > kMaxK1 || (lda & 3) || ((uintptr_t)a & 15) || ldctr < 3) return DET6D_EINVAL;
    if (c1 <= 0 || c1 > kMaxC || c2 <= 0 || c2 > kMaxC || c3 <= 0 || c3 > kMaxC3) return DET6D_EINVAL;
  }
  if (ldw1 < c1 || ldw2 < c2 || ldw3 < c3 || rows % (m * ns) || rows % 32) return DET6D_EINVAL;
  if (rows == 0) return DET6D_OK;
  ChainArgs g;
  g.rows = rows; g.n = n; g.m = m; g.ns = ns;
  g.a = a; g.lda = lda; g.idx = idx; g.ctr = ctr; g.ldctr = ldctr; g.cnt = cnt;
  g.w1 = w1; g.w2 = w2; g.w3 = w3; g.ldw1 = ldw1; g.ldw2 = ldw2; g.ldw3 = ldw3;
  g.s1 = s1; g.s2 = s2; g.s3 = s3;
  g.k1 = lda; g.c1 = c1; g.c2 = c2; g.c3 = c3;
  g.y = y; g.ldy = ldy; g.col0 = col0;
  g.hdr = nullptr; g.crow_p = nullptr; g.crow_c = nullptr;
  const int ntiles = rows / 32;
  if (wide) {   // one 512-thread workgroup per CU (weights fill most of its LDS), two waves per SIMD
    const size_t lds_bytes = sizeof(float) * ((size_t)70 * 64 + (size_t)66 * c2 + (size_t)c2 * 128);
    const int wb = det6d_divup(ntiles, 8) < 256 ? det6d_divup(ntiles, 8) : 256;
#define D6_WIDE(C2V, NSV)                                                                                              \
  do {                                                                                                                 \
    DET6D_MAX_DYNAMIC_LDS((mlp_chain_wide_kernel<C2V, NSV>), lds_bytes);                                               \
    hipLaunchKernelGGL((mlp_chain_wide_kernel<C2V, NSV>), dim3(wb), dim3(512), lds_bytes, (hipStream_t)stream, g);     \
  } while (0)
    if (c2 == 64 && ns == 16) D6_WIDE(64, 16);
    else if (c2 == 64) D6_WIDE(64, 32);
    else if (ns == 16) D6_WIDE(96, 16);
    else D6_WIDE(96, 32);
#undef D6_WIDE
    return det6d_check_launch("det6d_mlp_chain3");
  }
  int blocks = det6d_divup(ntiles, kChainWaves);
  if (blocks > 256 * 6) blocks = 256 * 6;     // persistent-ish: amortise the weight staging over many tiles
  static const bool use_lds_env = det6d_env_set("DET6D_CHAIN_LDS");
  const bool use_lds = use_lds_env || (ns == 16 && (m & 1));   // the register kernel pairs two centres of ONE batch per tile
  // register kernel: a grid of exactly one residency round (256 CUs x 4 SIMDs x 4 waves) so that every wave
  // walks the same number of tiles (1536 blocks left half the chip idle in the second round)
  static const int reg_blocks_env = det6d_env_int("DET6D_CHAIN_BLOCKS", 1024);
  const int reg_blocks = blocks < reg_blocks_env ? blocks : reg_blocks_env;
  if (!use_lds && lda == 4 && c1 == 16 && c2 == 16 && c3 == 32 && ns == 16)
    hipLaunchKernelGGL((mlp_chain_reg_kernel<16, 16, 32, 16>), dim3(reg_blocks), dim3(256), 0, (hipStream_t)stream, g);
  else if (!use_lds && lda == 4 && c1 == 16 && c2 == 16 && c3 == 32 && ns == 32)
    hipLaunchKernelGGL((mlp_chain_reg_kernel<16, 16, 32, 32>), dim3(reg_blocks), dim3(256), 0, (hipStream_t)stream, g);
  else if (!use_lds && lda == 4 && c1 == 32 && c2 == 32 && c3 == 64 && ns == 16)
    hipLaunchKernelGGL((mlp_chain_reg_kernel<32, 32, 64, 16>), dim3(reg_blocks), dim3(256), 0, (hipStream_t)stream, g);
  else if (!use_lds && lda == 4 && c1 == 32 && c2 == 32 && c3 == 64 && ns == 32)
    hipLaunchKernelGGL((mlp_chain_reg_kernel<32, 32, 64, 32>), dim3(reg_blocks), dim3(256), 0, (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL(mlp_chain_kernel, dim3(blocks), dim3(64 * kChainWaves), 0, (hipStream_t)stream, g);
  return det6d_check_launch("det6d_mlp_chain3");
}

// The same chains over a compact (ragged) row list: rows, classes and output centres come from det6d_compact_groups
// (hdr / crow_p / crow_c on the device), `capacity` only sizes the persistent grid.
DET6D_API int det6d_mlp_chain3_compact(int capacity, const int *hdr, const int *crow_p, const int *crow_c, const float *a,
                                       int lda, const float *ctr, int ldctr, const float *w1, int ldw1, const float *s1,
                                       int c1, const float *w2, int ldw2, const float *s2, int c2, const float *w3, int ldw3,
                                       const float *s3, int c3, float *y, int ldy, int col0, det6d_stream_t stream) {
  D6_GEMM_PRIO_HOST();
  if (capacity <= 0 || (capacity & 127) || !hdr || !crow_p || !crow_c || !a || !ctr || ldctr < 3 || !w1 || !w2 || !w3 || !s1 ||
      !s2 || !s3 || !y)
    return DET6D_EINVAL;
  if (ldw1 < c1 || ldw2 < c2 || ldw3 < c3) return DET6D_EINVAL;
  const bool wide = lda == 68 && c1 == 64 && (c2 == 64 || c2 == 96) && c3 == 128 && !(ldw1 & 3) && !(ldw2 & 3) && !(ldw3 & 3) &&
                    !(((uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)w3 | (uintptr_t)s1 | (uintptr_t)s2) & 15);
  const bool narrow = lda == 4 && !((uintptr_t)a & 15) && ((c1 == 16 && c2 == 16 && c3 == 32) || (c1 == 32 && c2 == 32 && c3 == 64));
  if (!wide && !narrow) return DET6D_EINVAL;
  ChainArgs g;
  g.rows = capacity; g.n = 0; g.m = 0; g.ns = 0;
  g.a = a; g.lda = lda; g.idx = nullptr; g.ctr = ctr; g.ldctr = ldctr; g.cnt = nullptr;
  g.w1 = w1; g.w2 = w2; g.w3 = w3; g.ldw1 = ldw1; g.ldw2 = ldw2; g.ldw3 = ldw3;
  g.s1 = s1; g.s2 = s2; g.s3 = s3;
  g.k1 = lda; g.c1 = c1; g.c2 = c2; g.c3 = c3;
  g.y = y; g.ldy = ldy; g.col0 = col0;
  g.hdr = hdr; g.crow_p = crow_p; g.crow_c = crow_c;
  const int ntiles = capacity / 32;
  if (wide) {
    const size_t lds_bytes = sizeof(float) * ((size_t)70 * 64 + (size_t)66 * c2 + (size_t)c2 * 128);
    const int wb = det6d_divup(ntiles, 8) < 256 ? det6d_divup(ntiles, 8) : 256;
    if (c2 == 64) {
      DET6D_MAX_DYNAMIC_LDS((mlp_chain_wide_kernel<64, 32, true>), lds_bytes);
      hipLaunchKernelGGL((mlp_chain_wide_kernel<64, 32, true>), dim3(wb), dim3(512), lds_bytes, (hipStream_t)stream, g);
    } else {
      DET6D_MAX_DYNAMIC_LDS((mlp_chain_wide_kernel<96, 32, true>), lds_bytes);
      hipLaunchKernelGGL((mlp_chain_wide_kernel<96, 32, true>), dim3(wb), dim3(512), lds_bytes, (hipStream_t)stream, g);
    }
    return det6d_check_launch("det6d_mlp_chain3_compact");
  }
  // grid: up to 2048 workgroups, each wave walks ceil(live tiles / waves) tiles (measured on 32-scene passes, SA1's wide
  // group alone / with the chip full: 768 workgroups 137 / 65 us, 1024: 115 / 64, 2048: 97 / 65, 4096: 87 / 70; ray-cast
  // scenes 578 / 205, 507 / 205, 366 / 201, 299 / 206)
  int blocks = det6d_divup(ntiles, 4);
  static const int cap = det6d_env_int("DET6D_CHAIN_BLOCKS", 2048);   // experiments build only
  if (blocks > cap) blocks = cap;
  if (c1 == 16)
    hipLaunchKernelGGL((mlp_chain_reg_kernel<16, 16, 32, 32, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL((mlp_chain_reg_kernel<32, 32, 64, 32, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
  return det6d_check_launch("det6d_mlp_chain3_compact");
}
