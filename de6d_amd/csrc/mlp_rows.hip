// mlp_rows.hip — short stacks of PLAIN pointwise layers over few rows in ONE launch: the aggregation + confidence
// chain of an SA layer (pointnet2_modules.py:580-607: Conv1d/BN/ReLU x 1 -> Conv1d/BN/ReLU -> Conv1d), the vote FC of the
// head (point_head_box6d_vote.py:33-45,815-818) and its cls / reg towers (:157-169, two chains over the same input).
// As separate det6d_linear launches these are 12 of the 24 GEMM-family launches of a pass and each is latency-bound
// (2 048 .. 32 768 rows, 0.0-0.5 GFLOP, 6-21 us on an idle chip); here a 32-row tile goes through the whole stack with
// its activations in LDS (row-major, odd stride: conflict-free MFMA A fragments) and the weights read straight from L2
// into the B fragments, two 16-k-step blocks ahead, the first two requested before the barrier that completes the layer's
// input — the scheme of mlp_group.hip at run-time widths.
// Every output is the same ascending-k fma chain as det6d_linear (+ shift, activation): bit-identical.
#include "common.h"

namespace {

D6_GEMM_PRIO_DECL

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4r __attribute__((ext_vector_type(4)));

constexpr int kMaxLayers = 4;

struct RowsArgs {
  int rows;
  const float *x; int ldx; int xcol0; int k0;     // input: columns [xcol0, xcol0 + k0) of x (rows, ldx)
  int wa, wb;                                     // LDS row widths: XA holds the input and the outputs of odd layers, XB the outputs of even layers
  int vec4;                                       // input rows are 16-byte aligned and k0 % 4 == 0
  int fits32;                                     // every output buffer is addressable with 32-bit byte offsets
  int kchunk;                                     // columns of the input held in LDS at a time (== k0: all of them)
  int nlayers[2];
  det6d_rows_layer layers[2][kMaxLayers];         // chain c = blockIdx.y
};

// Epilogue of one 32 x 32 accumulator tile of a plain layer: + shift, activation, the tile into the next layer's LDS image
// (`Y`, row stride LDY; nullptr for the last layer) and / or into the layer's output rows.  Round 6: the per-element form
// (`if (!last && cok) .. if (L.out && cok && r < rows) ..` with a 64-bit address per element) cost ~25 scalar / vector
// instructions per accumulator register — several times the MFMAs of a narrow layer; now the activation and store switches
// are wave-uniform branches around whole loops, and an INTERIOR tile (all 32 rows and all 32 columns live, 32-bit byte
// offsets) stores through the buffer path with the row of every accumulator register as a scalar offset.
__device__ __forceinline__ void rows_epilogue(f32x16 &acc, const det6d_rows_layer &L, const float sh, const int col, float *__restrict__ Y,
                                              const int LDY, const int rb_row0, const int kh, const int row_g0, const int rows,
                                              const bool fits32) {
  const bool cok = col < L.n;
  if (L.act == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = d6_relu(acc[e] + sh);
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = acc[e] + sh;
  }
  if (Y != nullptr && cok) {
    float *yw = Y + (rb_row0 + 4 * kh) * LDY + col;
#pragma unroll
    for (int e = 0; e < 16; ++e) yw[((e & 3) + 8 * (e >> 2)) * LDY] = acc[e];
  }
  if (L.out == nullptr) return;
  const int c0 = col & ~31;
  const bool interior = fits32 && row_g0 + rb_row0 + 32 <= rows && c0 + 32 <= L.n;      // wave-uniform
  if (interior) {
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)(L.out + L.ocol0), 0, 0xffffffff, 0x00020000);
    const int ldo4 = L.ldo * 4;
    const uint32_t voff = (uint32_t)(row_g0 + rb_row0 + 4 * kh) * (uint32_t)ldo4 + (uint32_t)col * 4u;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float v = acc[e];
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), srd, voff, ((e & 3) + 8 * (e >> 2)) * ldo4, 0);
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int r = row_g0 + rb_row0 + (e & 3) + 8 * (e >> 2) + 4 * kh;
    const float v = acc[e];
    if (cok && r < rows) L.out[(size_t)r * L.ldo + L.ocol0 + col] = v;
  }
}

// RB = 32-row blocks per tile.  RB = 2 (narrow stacks: at most two column tiles per layer, e.g. [96 -> 64 -> 32 -> 1]): the
// work items of a layer are (row block, column tile) pairs, so that a layer of two column tiles keeps all four waves busy
// and a layer of one keeps two, instead of two and one.
template <bool CHUNKED, int RB>
__global__ __launch_bounds__(256) void mlp_rows_kernel(const RowsArgs g) {
  D6_GEMM_PRIO_APPLY();
  static_assert(!(CHUNKED && RB != 1), "the K-chunked first layer keeps one accumulator per wave");
  constexpr int TR = 32 * RB;                     // rows per tile
  extern __shared__ float lds[];
  const int LDA = g.wa + 1, LDB = g.wb + 1;       // odd strides: conflict-free A fragments
  float *XA = lds, *XB = lds + TR * LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int chain = blockIdx.y;
  const int nl = g.nlayers[chain];
  const int ntiles_rows = (g.rows + TR - 1) / TR;
  constexpr int TPR = 8 / RB;                     // threads per row of the input tile
  const int lrow = tid / TPR, lq = tid % TPR;
  // next-tile input prefetch (whole-input stacks with 16-byte rows whose float4 deal evenly over the row's threads)
  constexpr int kPre = 8;
  const int nv = g.k0 / (4 * TPR);
  const bool pre = !CHUNKED && g.vec4 && nv * 4 * TPR == g.k0 && nv <= kPre;
  f32x4r nxt[kPre];
  auto gfetch = [&](const int t) {
    const int r = t * TR + lrow;
    const float *src = g.x + (size_t)(r < g.rows ? r : 0) * g.ldx + g.xcol0 + 4 * lq;
#pragma unroll
    for (int i = 0; i < kPre; ++i)
      if (i < nv) {
        nxt[i] = *reinterpret_cast<const f32x4r *>(src + 4 * TPR * i);
        if (r >= g.rows) nxt[i] = f32x4r{0.f, 0.f, 0.f, 0.f};
      }
  };
  if (pre && (int)blockIdx.x < ntiles_rows) gfetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles_rows; tile += gridDim.x) {
    // ---- input tile (columns [c0, c0 + kchunk)) -> XA (rows past the end: zeros); 8 threads per row, 16 bytes each where
    // the rows allow it ----
    auto load_input = [&](const int c0) {
      const int kw = CHUNKED ? g.kchunk : g.k0;
      const int r = tile * TR + lrow;
      const float *src = g.x + (size_t)(r < g.rows ? r : 0) * g.ldx + g.xcol0 + c0;
      float *dst = XA + lrow * LDA;
      if (g.vec4) {
        for (int c = 4 * lq; c < kw; c += 4 * TPR) {
          const f32x4r v = *reinterpret_cast<const f32x4r *>(src + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[c + e] = r < g.rows ? v[e] : 0.f;
        }
      } else {
        for (int c = lq; c < kw; c += TPR) dst[c] = r < g.rows ? src[c] : 0.f;
      }
    };
    if (pre) {
      // (round 6) the tile's rows were requested a tile ago (registers): LDS image now, the NEXT tile's rows requested before
      // the first layer, in flight during all of this tile's layers
      float *dst = XA + lrow * LDA;
#pragma unroll
      for (int i = 0; i < kPre; ++i)
        if (i < nv) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[4 * lq + 4 * TPR * i + e] = nxt[i][e];
        }
      if (tile + (int)gridDim.x < ntiles_rows) gfetch(tile + gridDim.x);
    } else {
      load_input(0);
    }
    for (int l = 0; l < nl; ++l) {
      const det6d_rows_layer &L = g.layers[chain][l];
      const float *X = (l & 1) ? XB : XA;
      float *Y = (l & 1) ? XA : XB;
      const int LD = (l & 1) ? LDB : LDA, LDY = (l & 1) ? LDA : LDB;
      const bool last = l == nl - 1;
      const __amdgpu_buffer_rsrc_t srd =
          __builtin_amdgcn_make_buffer_rsrc((void *)(L.w + (size_t)L.wrow0 * L.ldw), 0, (unsigned)((size_t)L.k * L.ldw * 4), 0x00020000);
      const int ldw_bytes = L.ldw * 4;
      const int ncol_tiles = (L.n + 31) >> 5;
      const int nblk = L.k >> 5;                      // blocks of 16 k-steps (k is a multiple of 32)
      // B fragments: ring of three register sets of 16 k-steps, two blocks ahead of their use.  Every fetch is
      // unconditional (past the end it re-reads the last block): behind a conditional fetch the compiler's s_waitcnt
      // vmcnt accounting assumes the shorter queue and drains the ring.  The first two blocks of a wave's first column
      // tile are requested BEFORE the barrier that completes the layer's input in LDS (weights do not depend on it).
      float bs[3][16];
      auto fetch = [&](float (&b)[16], uint32_t voff, int blk) {
        const int bb = blk < nblk ? blk : nblk - 1;
#pragma unroll
        for (int u = 0; u < 16; ++u)
          b[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, voff, 2 * (bb * 16 + u) * ldw_bytes, 0));
      };
      if (CHUNKED && l == 0) {
        // ---- wide input (the head's towers: 512 columns), at most four column tiles: the input passes through LDS in
        // chunks of kchunk columns = K-chunks of this layer in ascending order, every wave keeps the accumulator of ITS
        // column tile across the chunks (the same ascending-k chain); 49 instead of 82 KB of LDS: three workgroups per CU
        const int cblk = g.kchunk >> 5;
        const bool mine = wave < ncol_tiles;
        const int col = 32 * wave + l31;
        const uint32_t voff = (uint32_t)(kh * L.ldw + col) * 4u;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        for (int b0 = 0; b0 < nblk; b0 += cblk) {
          if (b0 > 0) {
            __syncthreads();                    // every wave is done with the previous chunk
            load_input(32 * b0);
          }
          if (mine) {
            fetch(bs[0], voff, b0);
            fetch(bs[1], voff, b0 + 1);
          }
          __syncthreads();
          if (mine) {
            auto compute = [&](const float (&b)[16], int blk) {
              const float *xa = X + l31 * LD + 32 * (blk - b0) + kh;
              float a[16];
#pragma unroll
              for (int u = 0; u < 16; ++u) a[u] = xa[2 * u];
#pragma unroll
              for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
            };
            const int bend = b0 + cblk;
            int blk = b0;
#pragma unroll 1
            for (; blk + 3 <= bend; blk += 3) {
              fetch(bs[2], voff, blk + 2);
              compute(bs[0], blk);
              fetch(bs[0], voff, blk + 3);
              compute(bs[1], blk + 1);
              fetch(bs[1], voff, blk + 4);
              compute(bs[2], blk + 2);
            }
            if (blk < bend) compute(bs[0], blk);
            if (blk + 1 < bend) compute(bs[1], blk + 1);
          }
        }
        if (mine) {
          const float sh = (col < L.n && L.shift) ? L.shift[col] : 0.f;
          rows_epilogue(acc, L, sh, col, last ? nullptr : Y, LDY, 0, kh, tile * 32, g.rows, g.fits32 != 0);
        }
        continue;
      }
      // work items: (row block rb, column tile j), item = RB * j + rb; wave w takes items w, w + 4, ...
      const int nitems = RB * ncol_tiles;
      if (wave < nitems) {
        const uint32_t voff0 = (uint32_t)(kh * L.ldw + 32 * (wave / RB) + l31) * 4u;
        fetch(bs[0], voff0, 0);
        fetch(bs[1], voff0, 1);
      }
      __syncthreads();
      for (int item = wave; item < nitems; item += 4) {
        const int j = item / RB, rb = item % RB;
        const int col = 32 * j + l31;
        const uint32_t voff = (uint32_t)(kh * L.ldw + col) * 4u;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        auto compute = [&](const float (&b)[16], int blk) {
          const float *xa = X + (32 * rb + l31) * LD + 32 * blk + kh;
          float a[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) a[u] = xa[2 * u];
#pragma unroll
          for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
        };
        if (item != wave) {
          fetch(bs[0], voff, 0);
          fetch(bs[1], voff, 1);
        }
        int blk = 0;
#pragma unroll 1
        for (; blk + 3 <= nblk; blk += 3) {
          fetch(bs[2], voff, blk + 2);
          compute(bs[0], blk);
          fetch(bs[0], voff, blk + 3);
          compute(bs[1], blk + 1);
          fetch(bs[1], voff, blk + 4);
          compute(bs[2], blk + 2);
        }
        if (blk < nblk) compute(bs[0], blk);           // nblk mod 3 blocks left: sets 0, 1 hold them
        if (blk + 1 < nblk) compute(bs[1], blk + 1);
        const float sh = (col < L.n && L.shift) ? L.shift[col] : 0.f;
        rows_epilogue(acc, L, sh, col, last ? nullptr : Y, LDY, 32 * rb, kh, tile * TR, g.rows, g.fits32 != 0);
      }
    }
    __syncthreads();     // the next tile's input overwrites XA, which the last layer may still be reading
  }
}

// ---- the SA1-shaped stack [K0 -> K1 -> K2 -> n <= 32] over MANY rows (the first level's aggregation + confidence chain:
// 4096 centres per scene), weights RESIDENT in registers --------------------------------------------------------------------
// Round 5 (scripts/r05/whatif_twice.py): the pass period pays for 97 % of this launch's stand-alone time, and the general kernel
// above spends it waiting, not computing — per 64-row tile 288 MFMAs (1.9 us of a CU's matrix pipes) inside ~20 us of input
// load -> barrier -> [weight fragments from L2 -> K loop -> barrier] x 3.  The whole stack's weights are 9 216 floats: every
// wave keeps the B fragments of ITS work items in registers for the life of the (persistent) workgroup — layer 0: K0 / 2
// registers for (row block = wave & 1, column tile = wave >> 1); waves 2, 3 the K1 / 2 of layer 1, waves 0, 1 the K2 / 2 of
// layer 2 — so a tile touches memory only for its input rows (prefetched into registers one tile ahead, double-buffered in
// LDS: three barriers per tile, none at its end) and its outputs.  Same MFMA sequence per output element as the general
// kernel (ascending k, + shift, activation): bit-identical.
// one work item of a layer: 32 rows (xa: this lane's row and k parity in LDS) x the wave's resident B fragments, NBLK blocks of
// 16 k-steps in ascending k
template <int NBLK, int NB>
__device__ __forceinline__ f32x16 rows_item(const float *xa, const float (&b)[NB]) {
  static_assert(16 * NBLK <= NB, "the fragment array holds the layer");
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    float a[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = xa[32 * blk + 2 * u];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[16 * blk + u], acc, 0, 0, 0);
  }
  return acc;
}

template <int K0, int K1, int K2>
__global__ __launch_bounds__(256) void mlp_rows_resident_kernel(const RowsArgs g) {
  D6_GEMM_PRIO_APPLY();
  constexpr int TR = 64, LDA = K0 + 1, LDB = K1 + 1, NV = K0 / 16;    // NV float4 per thread and input row (4 threads per row)
  static_assert(K0 % 32 == 0 && K1 % 32 == 0 && K2 % 32 == 0 && K2 <= K1 && K1 <= 64 && K2 <= K0, "SA1-shaped stacks only");
  extern __shared__ float lds[];
  float *XA0 = lds, *XA1 = lds + TR * LDA, *XB = lds + 2 * TR * LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const det6d_rows_layer &L0 = g.layers[0][0], &L1 = g.layers[0][1], &L2 = g.layers[0][2];
  const int rb = wave & 1, j0 = wave >> 1;
  const bool second = wave >= 2;                  // waves 2, 3 run layer 1, waves 0, 1 layer 2 (row block = wave & 1 in both)
  const det6d_rows_layer &LX = second ? L1 : L2;
  float b0[K0 / 2], bx[K1 / 2];
  {
    const __amdgpu_buffer_rsrc_t srd0 =
        __builtin_amdgcn_make_buffer_rsrc((void *)(L0.w + (size_t)L0.wrow0 * L0.ldw), 0, (unsigned)((size_t)L0.k * L0.ldw * 4), 0x00020000);
    const uint32_t voff0 = (uint32_t)(kh * L0.ldw + 32 * j0 + l31) * 4u;
#pragma unroll
    for (int u = 0; u < K0 / 2; ++u) b0[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd0, voff0, 2 * u * L0.ldw * 4, 0));
    const __amdgpu_buffer_rsrc_t srdx =
        __builtin_amdgcn_make_buffer_rsrc((void *)(LX.w + (size_t)LX.wrow0 * LX.ldw), 0, (unsigned)((size_t)LX.k * LX.ldw * 4), 0x00020000);
    const uint32_t voffx = (uint32_t)(kh * LX.ldw + l31) * 4u;
    const int kx = second ? K1 / 2 : K2 / 2;
#pragma unroll
    for (int u = 0; u < K1 / 2; ++u)
      bx[u] = u < kx ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srdx, voffx, 2 * u * LX.ldw * 4, 0)) : 0.f;
  }
  const int col0 = 32 * j0 + l31;
  const bool cok0 = col0 < L0.n, cokx = l31 < LX.n;
  const float sh0 = (cok0 && L0.shift) ? L0.shift[col0] : 0.f;
  const float shx = (cokx && LX.shift) ? LX.shift[l31] : 0.f;
  const int ntiles = (g.rows + TR - 1) / TR;
  const int lrow = tid >> 2, lq = tid & 3;
  f32x4r nxt[NV];
  auto gload = [&](const int tile) {
    const int r = tile * TR + lrow;
    const float *src = g.x + (size_t)(r < g.rows ? r : 0) * g.ldx + g.xcol0 + 4 * lq;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      nxt[i] = *reinterpret_cast<const f32x4r *>(src + 16 * i);
      if (r >= g.rows) nxt[i] = f32x4r{0.f, 0.f, 0.f, 0.f};
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) gload(tile);
  for (int buf = 0; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    float *XA = buf ? XA1 : XA0;
    {
      float *dst = XA + lrow * LDA + 4 * lq;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[16 * i + e] = nxt[i][e];
    }
    if (tile + (int)gridDim.x < ntiles) gload(tile + gridDim.x);     // the next tile's rows: in flight during this tile's layers
    __syncthreads();                                // the input tile is complete
    {
      const f32x16 acc = rows_item<K0 / 32>(XA + (32 * rb + l31) * LDA + kh, b0);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 32 * rb + (e & 3) + 8 * (e >> 2) + 4 * kh;
        float v = acc[e] + sh0;
        if (L0.act == 1) v = d6_relu(v);
        if (cok0) XB[row * LDB + col0] = v;
        const int r = tile * TR + row;
        if (L0.out && cok0 && r < g.rows) L0.out[(size_t)r * L0.ldo + L0.ocol0 + col0] = v;
      }
    }
    __syncthreads();                                // layer 0's output is complete (and the input tile is dead)
    if (second) {
      const f32x16 acc = rows_item<K1 / 32>(XB + (32 * rb + l31) * LDB + kh, bx);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 32 * rb + (e & 3) + 8 * (e >> 2) + 4 * kh;
        float v = acc[e] + shx;
        if (L1.act == 1) v = d6_relu(v);
        if (cokx) XA[row * LDA + l31] = v;
        const int r = tile * TR + row;
        if (L1.out && cokx && r < g.rows) L1.out[(size_t)r * L1.ldo + L1.ocol0 + l31] = v;
      }
    }
    __syncthreads();                                // layer 1's output is complete
    if (!second) {
      const f32x16 acc = rows_item<K2 / 32>(XA + (32 * rb + l31) * LDA + kh, bx);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 32 * rb + (e & 3) + 8 * (e >> 2) + 4 * kh;
        float v = acc[e] + shx;
        if (L2.act == 1) v = d6_relu(v);
        const int r = tile * TR + row;
        if (L2.out && cokx && r < g.rows) L2.out[(size_t)r * L2.ldo + L2.ocol0 + l31] = v;
      }
    }
    // no barrier here: the next tile's input goes to the OTHER input buffer, and XB is rewritten only behind the next tile's
    // first barrier, which waves 0, 1 reach after this layer
  }
}


// ---- round 6: the same SA1-shaped stack with WAVE-PRIVATE 32-row tiles -------------------------------------------------------
// mlp_rows_resident_kernel above still crosses three workgroup barriers per 64-row tile with its four waves specialised by
// layer (layer 1 runs on two waves, layer 2 on the other two while the rest wait): 98 us per 80-scene launch against a matrix
// floor of 38.  Here a wave owns a 32-row tile through all three layers and never waits for another wave: eight waves per CU
// (two workgroups of four) sit at eight different points of their tiles, so the matrix pipes see MFMAs from one wave while the
// other loads, stores or reads LDS.  Layer 0's B fragments (both column tiles, K0 registers) stay in registers for the life
// of the persistent wave, the small layers' weights in LDS (staged once per workgroup: the only workgroup barrier), the next
// tile's input rows are requested right after the current tile's have been written to LDS, and ONE wave-private LDS buffer of
// 32 x (K0 + 1) floats holds in turn the input, layer 0's output and layer 1's output (a layer's accumulators stay in registers
// until its last A fragment has been read; LDS serves a wave's instructions in order).  Same MFMA chain per output element:
// bit-identical to the other two kernels.
template <int K0, int K1, int K2>
__global__ __launch_bounds__(256, 2) void mlp_rows_wave_kernel(const RowsArgs g) {
  D6_GEMM_PRIO_APPLY();
  static_assert(K0 % 32 == 0 && K1 == 64 && K2 == 32, "SA1-shaped stacks: two column tiles, then one, then one");
  constexpr int LD0 = K0 + 1, LD1 = K1 + 1, LD2 = K2 + 1;
  constexpr int Q = K0 / 4;                          // float4 per input row
  // input tile = 4 blocks of 8 rows; a block's 8 Q float4 are dealt lane + 64 j (j < NJ): the (row, quad) a lane holds is the
  // same in every block, so its byte offsets are computed ONCE and a block / a tile only moves a scalar offset
  constexpr int NJ = 8 * Q / 64;
  static_assert(8 * Q % 64 == 0, "eight input rows fill whole wave-instructions");
  extern __shared__ float lds[];
  float *W1s = lds;                                  // K1 x 32 (layer 1's weights, k-major)
  float *W2s = lds + K1 * 32;                        // K2 x 32 (layer 2's, columns >= n zero)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  float *X = lds + (K1 + K2) * 32 + wave * (32 * LD0);      // wave-private
  const det6d_rows_layer &L0 = g.layers[0][0], &L1 = g.layers[0][1], &L2 = g.layers[0][2];
  for (int t = tid; t < K1 * 32; t += 256) {
    const int k = t >> 5, c = t & 31;
    W1s[t] = c < L1.n ? L1.w[(size_t)(L1.wrow0 + k) * L1.ldw + c] : 0.f;
  }
  for (int t = tid; t < K2 * 32; t += 256) {
    const int k = t >> 5, c = t & 31;
    W2s[t] = c < L2.n ? L2.w[(size_t)(L2.wrow0 + k) * L2.ldw + c] : 0.f;
  }
  // layer 0's B fragments: k-step s, column tile j: W0[2 s + kh][32 j + l31]
  float b0[2][K0 / 2];
  {
    const __amdgpu_buffer_rsrc_t srd0 =
        __builtin_amdgcn_make_buffer_rsrc((void *)(L0.w + (size_t)L0.wrow0 * L0.ldw), 0, (unsigned)((size_t)L0.k * L0.ldw * 4), 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t voff = (uint32_t)(kh * L0.ldw + 32 * j + l31) * 4u;
#pragma unroll
      for (int u = 0; u < K0 / 2; ++u)
        b0[j][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd0, voff, 2 * u * L0.ldw * 4, 0));
    }
  }
  float sh0[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) sh0[j] = L0.shift ? L0.shift[32 * j + l31] : 0.f;
  const bool cok1 = l31 < L1.n, cok2 = l31 < L2.n;
  const float sh1 = (cok1 && L1.shift) ? L1.shift[l31] : 0.f;
  const float sh2 = (cok2 && L2.shift) ? L2.shift[l31] : 0.f;
  __syncthreads();                                   // W1s / W2s staged: the only workgroup barrier

  // every tile is whole (the launcher asks for rows % 32 == 0) and every buffer is addressed with 32-bit byte offsets
  const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc((void *)(g.x + g.xcol0), 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_y0 = __builtin_amdgcn_make_buffer_rsrc((void *)(L0.out ? L0.out + L0.ocol0 : g.x), 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_y1 = __builtin_amdgcn_make_buffer_rsrc((void *)(L1.out ? L1.out + L1.ocol0 : g.x), 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_y2 = __builtin_amdgcn_make_buffer_rsrc((void *)(L2.out + L2.ocol0), 0, 0xffffffff, 0x00020000);
  uint32_t xin[NJ], xls[NJ];                         // per lane: byte offset of its float4 in the rows / in the LDS image
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int f = lane + 64 * j, row = f / Q, q = f % Q;
    xin[j] = (uint32_t)(row * g.ldx + 4 * q) * 4u;
    xls[j] = (uint32_t)(row * LD0 + 4 * q);
  }
  const bool out0 = L0.out != nullptr, out1 = L1.out != nullptr;
  const bool relu0 = L0.act == 1, relu1 = L1.act == 1, relu2 = L2.act == 1;
  const int ldo0 = L0.ldo * 4, ldo1 = L1.ldo * 4, ldo2 = L2.ldo * 4, ldx8 = g.ldx * 32;     // bytes (ldx8: eight input rows)

  const int ntiles = g.rows / 32;
  const int stride = gridDim.x * 4;
  f32x4r nxt[4][NJ];
  auto gload = [&](const int tile) {
    const int s0 = tile * 4 * ldx8;
#pragma unroll
    for (int bq = 0; bq < 4; ++bq)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        nxt[bq][j] = __builtin_bit_cast(f32x4r, __builtin_amdgcn_raw_buffer_load_b128(srd_x, xin[j], s0 + bq * ldx8, 0));
  };
  int tile = blockIdx.x * 4 + wave;
  if (tile < ntiles) gload(tile);
  for (; tile < ntiles; tile += stride) {
    // ---- input rows -> X (row stride LD0); the previous tile's last reads of X were issued before these writes ----
#pragma unroll
    for (int bq = 0; bq < 4; ++bq)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float *dst = X + xls[j] + bq * 8 * LD0;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = nxt[bq][j][e];
      }
    if (tile + stride < ntiles) gload(tile + stride);            // in flight during this tile's three layers
    __builtin_amdgcn_s_waitcnt(0xC07F);                          // lgkmcnt(0): the tile is in LDS (this wave wrote all of it)
    __builtin_amdgcn_wave_barrier();
    // ---- layer 0: two column tiles share every A fragment ----
    f32x16 a0[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) a0[j][e] = 0.f;
    {
      const float *xa = X + l31 * LD0 + kh;
#pragma unroll
      for (int blk = 0; blk < K0 / 32; ++blk) {
        float a[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] = xa[32 * blk + 2 * u];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          a0[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b0[0][16 * blk + u], a0[0], 0, 0, 0);
          a0[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b0[1][16 * blk + u], a0[1], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();                             // (every read of the input is issued: X may be rewritten)
    const uint32_t vrow = (uint32_t)(tile * 32 + 4 * kh);        // first row of this lane's accumulator registers
    // (activation and store switches are wave-uniform: one scalar branch per tile each, none per element)
    if (relu0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) a0[j][e] = d6_relu(a0[j][e] + sh0[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) a0[j][e] = a0[j][e] + sh0[j];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float *xw = X + (4 * kh) * LD1 + 32 * j + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) xw[((e & 3) + 8 * (e >> 2)) * LD1] = a0[j][e];
    }
    if (out0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t voff = vrow * (uint32_t)ldo0 + (uint32_t)(32 * j + l31) * 4u;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          // (a local copy: __builtin_bit_cast straight on the vector ELEMENT stored element 0 sixteen times — seen in the ISA)
          const float v = a0[j][e];
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), srd_y0, voff, ((e & 3) + 8 * (e >> 2)) * ldo0, 0);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // ---- layer 1 ----
    f32x16 a1;
#pragma unroll
    for (int e = 0; e < 16; ++e) a1[e] = 0.f;
    {
      const float *xa = X + l31 * LD1 + kh;
      const float *wb = W1s + kh * 32 + l31;
#pragma unroll
      for (int blk = 0; blk < K1 / 32; ++blk) {
        float a[16], b[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { a[u] = xa[32 * blk + 2 * u]; b[u] = wb[(32 * blk + 2 * u) * 32]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], a1, 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (relu1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) a1[e] = d6_relu(a1[e] + sh1);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) a1[e] = a1[e] + sh1;
    }
    {
      float *xw = X + (4 * kh) * LD2 + l31;          // (L1.n == K2 == 32: every column is a real one)
#pragma unroll
      for (int e = 0; e < 16; ++e) xw[((e & 3) + 8 * (e >> 2)) * LD2] = a1[e];
    }
    if (out1 && cok1) {
      const uint32_t voff = vrow * (uint32_t)ldo1 + (uint32_t)l31 * 4u;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = a1[e];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), srd_y1, voff, ((e & 3) + 8 * (e >> 2)) * ldo1, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // ---- layer 2 ----
    f32x16 a2;
#pragma unroll
    for (int e = 0; e < 16; ++e) a2[e] = 0.f;
    {
      const float *xa = X + l31 * LD2 + kh;
      const float *wb = W2s + kh * 32 + l31;
      float a[16], b[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { a[u] = xa[2 * u]; b[u] = wb[(2 * u) * 32]; }
#pragma unroll
      for (int u = 0; u < 16; ++u) a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], a2, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    if (relu2) {
#pragma unroll
      for (int e = 0; e < 16; ++e) a2[e] = d6_relu(a2[e] + sh2);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) a2[e] = a2[e] + sh2;
    }
    if (cok2) {
      const uint32_t voff = vrow * (uint32_t)ldo2 + (uint32_t)l31 * 4u;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = a2[e];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), srd_y2, voff, ((e & 3) + 8 * (e >> 2)) * ldo2, 0);
      }
    }
  }
}

}  // namespace

// Validates a stack description and derives the LDS plan: k0 (input width), wa / wb (row widths of the two activation
// buffers), kchunk (columns of the input held in LDS at a time).  Shared by det6d_mlp_rows and det6d_mlp_rows_supported.
static int rows_plan(int nchains, const int *nlayers, const det6d_rows_layer *layers, RowsArgs &g) {
  if (nchains < 1 || nchains > 2 || !nlayers || !layers) return DET6D_EINVAL;
  int wa = 0, wb = 0, k0 = -1, off = 0;
  for (int c = 0; c < 2; ++c) g.nlayers[c] = 0;
  for (int c = 0; c < nchains; ++c) {
    const int nl = nlayers[c];
    if (nl < 1 || nl > kMaxLayers) return DET6D_EINVAL;
    g.nlayers[c] = nl;
    int kin = -1;
    for (int l = 0; l < nl; ++l) {
      const det6d_rows_layer &L = layers[off + l];
      if (!L.w || L.k <= 0 || (L.k & 31) || L.n <= 0 || L.ldw < L.n || L.wrow0 < 0) return DET6D_EINVAL;
      if (l == 0) {
        if (k0 >= 0 && L.k != k0) return DET6D_EINVAL;      // both chains read the same input tile
        k0 = L.k;
      } else if (L.k != kin) {
        return DET6D_EINVAL;                                // a hidden layer's width is the next layer's chain length
      }
      if (l < nl - 1 && (L.n & 31)) return DET6D_EINVAL;
      if (l == nl - 1 && !L.out) return DET6D_EINVAL;
      if (L.out && (L.ldo < L.ocol0 + L.n)) return DET6D_EINVAL;
      if ((size_t)L.k * L.ldw * 4 >= 0xfff00000ull) return DET6D_EINVAL;
      if (l & 1) { if (L.k > wb) wb = L.k; } else { if (L.k > wa) wa = L.k; }     // layer l reads XA (even l) / XB (odd l)
      kin = L.n;
      g.layers[c][l] = L;
    }
    off += nl;
  }
  g.k0 = k0; g.wa = wa; g.wb = wb > 0 ? wb : 1;
  g.kchunk = k0;
  if (k0 >= 512 && (k0 % 256) == 0) {      // a wide input whose first layers have at most four column tiles: K-chunks of 256
    bool narrow = true;
    int wa_rest = 256, o = 0;
    for (int c = 0; c < nchains; ++c) {
      if (layers[o].n > 128) narrow = false;
      for (int l = 2; l < nlayers[c]; l += 2) if (layers[o + l].k > wa_rest) wa_rest = layers[o + l].k;
      o += nlayers[c];
    }
    if (narrow) { g.kchunk = 256; g.wa = wa_rest; }
  }
  // one 32-row tile's two activation buffers must fit the CU's 160 KB (64-row tiles are chosen only for narrow stacks)
  if (sizeof(float) * 32 * ((size_t)(g.wa + 1) + (size_t)(g.wb + 1)) > 160 * 1024) return DET6D_EINVAL;
  return DET6D_OK;
}

// 1 when det6d_mlp_rows accepts this stack (widths, chain structure, LDS), else 0: the host asks BEFORE it routes a stack of
// plain layers here instead of through one det6d_linear per layer (e.g. a [1024 -> 1024 -> ..] tower does not fit).
DET6D_API int det6d_mlp_rows_supported(int nchains, const int *nlayers, const det6d_rows_layer *layers) {
  RowsArgs g;
  return rows_plan(nchains, nlayers, layers, g) == DET6D_OK ? 1 : 0;
}

DET6D_API int det6d_mlp_rows(int rows, const float *x, int ldx, int xcol0, int nchains, const int *nlayers,
                             const det6d_rows_layer *layers, det6d_stream_t stream) {
  D6_GEMM_PRIO_HOST();
  if (rows < 0 || !x || ldx <= 0 || xcol0 < 0) return DET6D_EINVAL;
  RowsArgs g;
  if (rows_plan(nchains, nlayers, layers, g) != DET6D_OK) return DET6D_EINVAL;
  g.rows = rows; g.x = x; g.ldx = ldx; g.xcol0 = xcol0;
  const int k0 = g.k0;
  if (xcol0 + k0 > ldx) return DET6D_EINVAL;
  g.vec4 = ((k0 & 3) == 0 && (ldx & 3) == 0 && (xcol0 & 3) == 0 && (((uintptr_t)x) & 15) == 0) ? 1 : 0;
  if (rows == 0) return DET6D_OK;
  // the first level's stack ([96 -> 64 -> 32 -> 1] over 4096 centres per scene): weights resident in registers
  static const int resident_env = det6d_env_int("DET6D_ROWS_RESIDENT", 2);
  // DET6D_ROWS_RESIDENT (knobs build): 2 (default) = wave-private tiles (round 6), 1 = the four-wave resident kernel of round 5,
  // 0 = the general kernel
  // (32-bit byte offsets into the input and every output: the buffer-store epilogues and the wave-private kernel)
  bool fits32 = (size_t)rows * ldx * 4 < 0xfff00000ull;
  {
    int nl_all = 0;
    for (int c = 0; c < nchains; ++c) nl_all += nlayers[c];
    for (int l = 0; l < nl_all; ++l) fits32 = fits32 && (!layers[l].out || (size_t)rows * layers[l].ldo * 4 < 0xfff00000ull);
  }
  g.fits32 = fits32 ? 1 : 0;
  // (the wave-private kernel forms its scalar byte offsets in signed 32-bit arithmetic: half the range)
  const bool fits31 = fits32 && (size_t)rows * ldx * 4 < 0x7ff00000ull;
  if (resident_env == 2 && nchains == 1 && nlayers[0] == 3 && g.kchunk == g.k0 && g.vec4 && rows >= 16384 && (rows & 31) == 0 && fits31 &&
      layers[0].k == 96 && layers[0].n == 64 && layers[1].n == 32 && layers[2].n <= 32 && layers[0].ldw >= 64 && layers[1].ldw >= 32) {
    constexpr size_t lds_wave = sizeof(float) * ((64 + 32) * 32 + 4 * 32 * 97);
    DET6D_MAX_DYNAMIC_LDS((mlp_rows_wave_kernel<96, 64, 32>), lds_wave);
    int blocks = ((rows + 31) / 32 + 3) / 4;
    if (blocks > 512) blocks = 512;                  // two workgroups of four independent waves per CU, persistent over the tiles
    hipLaunchKernelGGL((mlp_rows_wave_kernel<96, 64, 32>), dim3(blocks), dim3(256), lds_wave, (hipStream_t)stream, g);
    return det6d_check_launch("det6d_mlp_rows");
  }
  if (resident_env && nchains == 1 && nlayers[0] == 3 && g.kchunk == g.k0 && g.vec4 && rows >= 16384 && layers[0].k == 96 &&
      layers[0].n == 64 && layers[1].n == 32 && layers[2].n <= 32 && layers[0].ldw >= 64 && layers[1].ldw >= 32) {
    constexpr size_t lds_resident = sizeof(float) * (2 * 64 * 97 + 64 * 65);
    DET6D_MAX_DYNAMIC_LDS((mlp_rows_resident_kernel<96, 64, 32>), lds_resident);
    int blocks = (rows + 63) / 64;
    if (blocks > 512) blocks = 512;                  // two workgroups per CU (66 KB of LDS each), persistent over the tiles
    hipLaunchKernelGGL((mlp_rows_resident_kernel<96, 64, 32>), dim3(blocks), dim3(256), lds_resident, (hipStream_t)stream, g);
    return det6d_check_launch("det6d_mlp_rows");
  }
  // narrow stacks (every layer at most two column tiles, single chain, many rows): 64-row tiles
  int max_tiles = 0;
  for (int c = 0, o = 0; c < nchains; o += nlayers[c], ++c)
    for (int l = 0; l < nlayers[c]; ++l) max_tiles = max_tiles > (layers[o + l].n + 31) / 32 ? max_tiles : (layers[o + l].n + 31) / 32;
  // DET6D_ROWS_RB: 2 (default) = 64-row tiles for narrow stacks over >= 16384 rows, 3 = over any number of rows, 1 = never
  static const int rb_env = det6d_env_int("DET6D_ROWS_RB", 2);
  const int rb = (rb_env >= 2 && g.kchunk == g.k0 && max_tiles <= 2 && (rows >= 16384 || rb_env == 3)) ? 2 : 1;
  const size_t lds_bytes = sizeof(float) * 32 * rb * ((size_t)(g.wa + 1) + (size_t)(g.wb + 1));
  if (lds_bytes > 160 * 1024) return DET6D_EINVAL;
  static size_t attr_bytes = 0;
  if (lds_bytes > attr_bytes) {
    hipFuncSetAttribute((const void *)mlp_rows_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipFuncSetAttribute((const void *)mlp_rows_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipFuncSetAttribute((const void *)mlp_rows_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_bytes = lds_bytes;
  }
  int blocks = (rows + 32 * rb - 1) / (32 * rb);
  // persistent walk over the tiles by as many workgroups as the chip HOLDS at a time (LDS: 3 per CU for the SA stacks; 128
  // registers: at most 4): a grid of 1024 on 768 slots ran its last 256 workgroups on a third of the chip (80-scene passes,
  // SA1's stack: 768: 132 us, 1024: 153, 1536: 134, 2048: 139, 512: 161; scripts/r05/gpu_t29.sh).  DET6D_ROWS_BLOCKS:
  // experiments build only.
  int per_cu = (int)((160 * 1024) / lds_bytes);
  per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
  static const int blocks_env = det6d_env_int("DET6D_ROWS_BLOCKS", 0);
  const int blocks_cap = blocks_env > 0 ? blocks_env : 256 * per_cu / nchains;     // the grid is (blocks, chains)
  if (blocks > blocks_cap) blocks = blocks_cap;
  if (g.kchunk < g.k0)
    hipLaunchKernelGGL((mlp_rows_kernel<true, 1>), dim3(blocks, nchains), dim3(256), lds_bytes, (hipStream_t)stream, g);
  else if (rb == 2)
    hipLaunchKernelGGL((mlp_rows_kernel<false, 2>), dim3(blocks, nchains), dim3(256), lds_bytes, (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL((mlp_rows_kernel<false, 1>), dim3(blocks, nchains), dim3(256), lds_bytes, (hipStream_t)stream, g);
  return det6d_check_launch("det6d_mlp_rows");
}
