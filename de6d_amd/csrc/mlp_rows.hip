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
  int kchunk;                                     // columns of the input held in LDS at a time (== k0: all of them)
  int nlayers[2];
  det6d_rows_layer layers[2][kMaxLayers];         // chain c = blockIdx.y
};

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
    load_input(0);
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
          const bool cok = col < L.n;
          const float sh = (cok && L.shift) ? L.shift[col] : 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * kh;
            float v = acc[e] + sh;
            if (L.act == 1) v = d6_relu(v);
            if (!last && cok) Y[row * LDY + col] = v;
            const int r = tile * 32 + row;
            if (L.out && cok && r < g.rows) L.out[(size_t)r * L.ldo + L.ocol0 + col] = v;
          }
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
        const bool cok = col < L.n;
        const float sh = (cok && L.shift) ? L.shift[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = 32 * rb + (e & 3) + 8 * (e >> 2) + 4 * kh;
          float v = acc[e] + sh;
          if (L.act == 1) v = d6_relu(v);
          if (!last && cok) Y[row * LDY + col] = v;
          const int r = tile * TR + row;
          if (L.out && cok && r < g.rows) L.out[(size_t)r * L.ldo + L.ocol0 + col] = v;
        }
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
  static const int resident_env = det6d_env_int("DET6D_ROWS_RESIDENT", 1);      // experiments build: 0 = the general kernel
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
