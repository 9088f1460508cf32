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
};

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
template <int K, int TN, int LDX>
__device__ __forceinline__ void group_layer(const float *__restrict__ X, const __amdgpu_buffer_rsrc_t srd, const uint32_t voff,
                                            const int ldw_bytes, f32x16 (&acc)[TN], const int l31, const int kh) {
  constexpr int VW = TN >= 4 ? 4 : TN;
  constexpr int NV = TN / VW;         // loads per k-step
  constexpr int KS = K / 2;           // k-steps of two
  constexpr int UK = 32 / TN;         // k-steps per block
  constexpr int NB = KS / UK;         // blocks
  static_assert(KS % UK == 0 && NB >= 2, "block structure");
  typedef typename BVec<VW>::T bvec;
  bvec b0[UK][NV], b1[UK][NV], b2[UK][NV];
  auto fetch = [&](bvec (&b)[UK][NV], int blk) {
#pragma unroll
    for (int u = 0; u < UK; ++u)
#pragma unroll
      for (int v = 0; v < NV; ++v) b[u][v] = load_b<VW>(srd, voff, 2 * (blk * UK + u) * ldw_bytes + 128 * VW * v);
  };
  auto compute = [&](const bvec (&b)[UK][NV], int blk) {
    float a[UK];
    const float *xa = X + l31 * LDX + 2 * blk * UK + kh;
#pragma unroll
    for (int u = 0; u < UK; ++u) a[u] = xa[2 * u];
#pragma unroll
    for (int u = 0; u < UK; ++u)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bget<VW>(b[u][j / VW], j % VW), acc[j], 0, 0, 0);
  };
  // three register sets in rotation: while block i computes, blocks i+1 and i+2 are in flight
  fetch(b0, 0);
  fetch(b1, 1);
  int blk = 0;
#pragma unroll 1
  for (; blk + 3 <= NB; blk += 3) {
    if (blk + 2 < NB) fetch(b2, blk + 2);
    compute(b0, blk);
    if (blk + 3 < NB) fetch(b0, blk + 3);
    compute(b1, blk + 1);
    if (blk + 4 < NB) fetch(b1, blk + 4);
    compute(b2, blk + 2);
  }
  if (blk < NB) {             // NB mod 3 == 1 or 2 (NB is a power of two)
    if (blk + 2 < NB) fetch(b2, blk + 2);
    compute(b0, blk);
    if (blk + 1 < NB) compute(b1, blk + 1);
  }
}

template <int C1, int C2, int C3, bool COMPACT, int NW>
__global__ __launch_bounds__(64 * NW) void mlp_group_kernel(const GroupArgs g) {
  constexpr int LD1 = C1 + 1, LD2 = C2 + 1;
  constexpr int TN2 = C2 / (32 * NW), TN3 = C3 / (32 * NW);     // accumulator tiles per wave: a wave owns 1 / NW of every layer's columns
  static_assert(TN2 >= 1 && TN3 >= 1, "every wave needs at least one 32-column tile per layer");
  extern __shared__ float lds[];
  float *X1 = lds;
  float *X2 = lds + 32 * LD1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
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

  // layer 1 (expand): thread (row = tid / 8, q = tid % 8) produces columns 4q + 32 i of its row
  constexpr int EPR = 2 * NW;                     // threads per row in the expand phase
  const int erow = tid / EPR, eq = tid % EPR;

  for (int tile = blockIdx.x; tile < live_tiles; tile += gridDim.x) {
    // ---- layer 1: X1[row][c] = relu(fma(dz, W1[2][c], fma(dy, W1[1][c], fma(dx, W1[0][c], P[p][c]))) + s1[c]) ----
    {
      const int r = tile * 32 + erow;
      long long prow;
      int cj;
      bool real = true;
      if (COMPACT) {
        const int tag = g.crow_c[r];
        real = tag >= 0;
        cj = tag & 0x1fffffff;
        prow = g.crow_p[r];
      } else {
        cj = r / g.ns;
        prow = (long long)(cj / g.m) * g.n + g.idx[r];
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
    __syncthreads();
    // ---- layer 2: X2 = relu(X1 W2 + s2) ----
    {
      f32x16 acc[TN2];
#pragma unroll
      for (int j = 0; j < TN2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
      group_layer<C1, TN2, LD1>(X1, srd2, voff2, g.ldw2 * 4, acc, l31, kh);
#pragma unroll
      for (int j = 0; j < TN2; ++j) {
        float *xc = X2 + wave * (C2 / NW) + tile_col<TN2>(j, l31);
#pragma unroll
        for (int e = 0; e < 16; ++e) xc[((e & 3) + 8 * (e >> 2) + 4 * kh) * LD2] = d6_relu(acc[j][e] + sh2[j]);
      }
    }
    __syncthreads();
    // ---- layer 3 + pooling ----
    f32x16 acc[TN3];
#pragma unroll
    for (int j = 0; j < TN3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    group_layer<C2, TN3, LD2>(X2, srd3, voff3, g.ldw3 * 4, acc, l31, kh);
    if (COMPACT) {
      const int sc = g_class(tile * 32, h1, h2, h3, h4, h5);
      if (sc < 4) {       // classes 1, 2: every accumulator (pair) is a centre part of its own
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (sc == 2 && (e & 1)) continue;
          const int tag = g.crow_c[tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh];
          if (tag < 0) continue;
          float *dst = g.y + (size_t)(tag & 0x1fffffff) * g.ldy + g.col0 + wave * (C3 / NW);
#pragma unroll
          for (int j = 0; j < TN3; ++j) {
            const float raw = sc == 2 ? d6_vmax(acc[j][e], acc[j][e + 1 < 16 ? e + 1 : e]) : acc[j][e];
            g_store(dst + tile_col<TN3>(j, l31), (tag & 0x40000000) ? 0.f : d6_relu(raw + sh3[j]), tag);
          }
        }
      } else {
        int oc[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int rr = g_out_row(sc, qq, kh);
          oc[qq] = rr >= 0 ? g.crow_c[tile * 32 + rr] : -1;
        }
#pragma unroll
        for (int j = 0; j < TN3; ++j) {
          float q[4];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
            q[qq] = d6_vmax(d6_vmax(acc[j][4 * qq], acc[j][4 * qq + 1]), d6_vmax(acc[j][4 * qq + 2], acc[j][4 * qq + 3]));
          g_pool(q, sc);
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
            if (oc[qq] >= 0)
              g_store(g.y + (size_t)(oc[qq] & 0x1fffffff) * g.ldy + g.col0 + wave * (C3 / NW) + tile_col<TN3>(j, l31),
                      (oc[qq] & 0x40000000) ? 0.f : d6_relu(q[qq] + sh3[j]), oc[qq]);
        }
      }
    } else {
      // dense rows: a tile is one centre (nsample 32) or two (nsample 16); empty balls pool to 0
      const int c0 = g.ns == 32 ? tile : 2 * tile;
      const int cnt0 = g.cnt[c0], cnt1 = g.ns == 32 ? 0 : g.cnt[c0 + 1];
#pragma unroll
      for (int j = 0; j < TN3; ++j) {
        float q[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const float mq = d6_vmax(d6_vmax(acc[j][4 * qq], acc[j][4 * qq + 1]), d6_vmax(acc[j][4 * qq + 2], acc[j][4 * qq + 3]));
          q[qq] = d6_vmax(mq, __shfl_xor(mq, 32));
        }
        float *dst = g.y + g.col0 + wave * (C3 / NW) + tile_col<TN3>(j, l31);
        if (kh == 0) {
          if (g.ns == 32) {
            const float mx = d6_relu(d6_vmax(d6_vmax(q[0], q[1]), d6_vmax(q[2], q[3])) + sh3[j]);
            dst[(size_t)c0 * g.ldy] = cnt0 > 0 ? mx : 0.f;
          } else {
            dst[(size_t)c0 * g.ldy] = cnt0 > 0 ? d6_relu(d6_vmax(q[0], q[1]) + sh3[j]) : 0.f;
            dst[(size_t)(c0 + 1) * g.ldy] = cnt1 > 0 ? d6_relu(d6_vmax(q[2], q[3]) + sh3[j]) : 0.f;
          }
        }
      }
    }
    // no barrier here: the next tile's layer 1 writes X1, which every wave finished reading before the barrier above;
    // X2 is rewritten only after the next tile's first barrier, which no wave passes before it has left layer 3
  }
}

template <int C1, int C2, int C3, bool COMPACT, int NW>
int launch_group(const GroupArgs &g, hipStream_t stream) {
  const size_t lds_bytes = sizeof(float) * 32 * (size_t)(C1 + 1 + C2 + 1);
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
  hipLaunchKernelGGL((mlp_group_kernel<C1, C2, C3, COMPACT, NW>), dim3(blocks), dim3(64 * NW), lds_bytes, stream, g);
  return det6d_check_launch("det6d_mlp_group3");
}

}  // namespace

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
                               int ns, const int *cnt, const int *hdr, const int *crow_p, const int *crow_c, float *y, int ldy,
                               int col0, det6d_stream_t stream) {
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
  g.y = y; g.ldy = ldy; g.col0 = col0;
  hipStream_t s = (hipStream_t)stream;
  // waves per 32-row tile: 8 for the head's groups (two waves per SIMD from ONE workgroup: the 99 KB of LDS allow only one
  // workgroup per CU), 4 for the SA3 groups (several workgroups per CU); DET6D_GROUP_WAVES (experiments build) overrides
  static const int nw_env = det6d_env_int("DET6D_GROUP_WAVES", 0);
#define D6_GROUP(A, B, C, NWD)                                                                        \
  if (c1 == A && c2 == B && c3 == C) {                                                                \
    if ((nw_env ? nw_env : NWD) == 8 && B >= 256)                                                     \
      return compact ? launch_group<A, B, C, true, (B >= 256 ? 8 : 4)>(g, s) : launch_group<A, B, C, false, (B >= 256 ? 8 : 4)>(g, s); \
    return compact ? launch_group<A, B, C, true, 4>(g, s) : launch_group<A, B, C, false, 4>(g, s);    \
  }
  D6_GROUP(128, 128, 256, 4)
  D6_GROUP(128, 256, 256, 4)
  D6_GROUP(256, 256, 512, 8)
  D6_GROUP(256, 512, 1024, 8)
#undef D6_GROUP
  return DET6D_EINVAL;
}
