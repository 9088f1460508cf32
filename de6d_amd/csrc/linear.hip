// linear.hip — the pointwise (1x1 conv / FC) layers of the SA modules and the head as fp32 MFMA
// GEMMs with fused prologue (neighbour gather + centre subtraction) and epilogue
// (folded-BN shift, ReLU, empty-ball mask, max-pool over the neighbour axis).
//
// Replaces the unfused chain of _PointnetSAModuleFSBase.forward
// (core/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:462-494: grouping_operation x2,
//  cat, Conv2d, BatchNorm2d, ReLU, mask multiply, max_pool2d) and the Conv1d/BN/ReLU stacks of
// PointHeadBox6DVote (core/pcdet/models/dense_heads/point_head_box6d_vote.py:33-78,157-169).
//
// Arithmetic contract (what makes the result bit-identical to oracle/det6d_oracle.c):
//   out[r][c] = act( fma-chain_{k ascending}( A'[r][k] * W[k][c] ) + shift[c] )
// v_mfma_f32_32x32x2_f32 is exactly that chain: D = fma(a_k1,b_k1, fma(a_k0,b_k0, C)); each
// output lives in ONE accumulator for the whole K loop (no split-K, no reassociation).
//
// Tiling for gfx950: 128 x BN output tile per 256-thread workgroup (4 waves), BK = 16, operands
// staged k-major in LDS so that both MFMA fragments are conflict-free ds_read_b32
// (A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31]); next tile's global loads are issued
// before the MFMA block of the current one (register prefetch); 16.5 KB LDS -> several
// workgroups per CU hide each other's barriers.
#include "common.h"
#include <stdlib.h>

namespace {

D6_GEMM_PRIO_DECL

typedef float f32x16 __attribute__((ext_vector_type(16)));


__device__ __forceinline__ float relu_act(float v, int act) { return act == 1 ? (v > 0.f ? v : 0.f) : v; }

// BK: k-tile depth (16 or 32).  NBUF = 2 double-buffers the LDS tiles (one barrier per iteration).
// Both were measured again after the vector-ALU clean-up (same-call A/B on MI355X):
// BK = 32: 582 vs 539 us on 65536 x 512 x 1024; NBUF = 2: GEMM family 1.873 vs 1.840 ms.  Kept as knobs.
// The bare LDS-fed MFMA loop (scripts/hiptests/lds_mfma.hip) reaches 138-155 TF depending on the box; the
// largest layer runs at 123-129 TF in the model.
typedef float f32x4v __attribute__((ext_vector_type(4)));

// FAST: the steady-state slabs are fetched with buffer_load_dwordx4 from per-thread byte offsets computed once
// (rows / columns outside the problem are clamped to offset 0: they only feed outputs that are never stored)
// plus a scalar k offset, so the main loop carries no address arithmetic, compares or zero-fill moves.
// On gfx950 the fp32 MFMA shares the VALU: every vector ALU instruction in the loop is ~5 cycles taken from
// the matrix pipe (scripts/hiptests/mfma_valu_overlap.hip), and the predicated form had ~56 of them per slab.
template <int BM, int BN, int WMW, int WNW, int TM, int TN, int BK = 16, int NBUF = 1, int FASTLVL = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void linear_kernel(const det6d_linear_args g) {
  D6_GEMM_PRIO_APPLY();
  constexpr bool FAST = FASTLVL >= 2;      // predicate-free buffer_load main loop
  // PIPE (round 6): the steady-state slab as an explicit software pipeline.  The ISA of the FASTLVL = 2 loop showed the
  // scheduler SINKING the next slab's four buffer loads below the slab's MFMAs (they left right before the closing barrier, and
  // the next iteration opened with s_waitcnt vmcnt: a full L2 round trip exposed per slab, hidden only by the other workgroups
  // of the CU), and every k-step as ds_read -> s_waitcnt lgkmcnt(0) -> 4 MFMAs (the LDS latency exposed eight times per slab).
  // Now: the loads are pinned above the MFMA block (sched_barrier), and the fragments of k-step s + 1 are read while the MFMAs
  // of k-step s run (two fragment sets in registers).
  constexpr bool PIPE = FASTLVL >= 3;
  constexpr bool FAST_EPI = FASTLVL >= 1;  // predicate-free buffer_store epilogue on interior tiles (32-bit offsets)
  constexpr int LDA_S = BM + 2;  // +2 makes the transposing ds_write_b32 conflict-free (see below)
  constexpr int NA = BM / 64;    // A rows per thread
  constexpr int KU = BK / 16;    // k-quads per row per thread
  static_assert(WMW * WNW == 4, "4 waves");
  static_assert(32 * TM * WMW == BM, "row tiling");
  static_assert(32 * TN * WNW == BN, "col tiling");
  __shared__ float As_all[NBUF * BK * LDA_S];
  __shared__ float Bs_all[NBUF * BK * BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WNW, wn = wave % WNW;
  const int K = g.k, N = g.ncols;
  // compact (ragged) rows: the row count lives on the device (csrc/compact.hip), g.rows is the capacity the
  // grid was sized for; tiles past the live rows leave at once.  R is a multiple of 128 there.
  const int R = g.hdr ? g.hdr[0] : g.rows;
  // XCD-aware tile order (1-D grid): workgroups are dealt round-robin over the 8 XCDs, each with its
  // own L2.  All column tiles of a row tile get consecutive slots on ONE XCD, so the A rows they share
  // are fetched into that L2 once instead of once per column tile (speed only, never correctness).
  const int gm = (g.rows + BM - 1) / BM, gn = (N + BN - 1) / BN;
  int row_tile, col_tile;
  if (gn > 1 && (gm & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    row_tile = (slot / gn) * 8 + xcd;
    col_tile = slot % gn;
  } else {
    row_tile = blockIdx.x / gn;
    col_tile = blockIdx.x % gn;
  }
  const int row0 = row_tile * BM;
  const int colb = col_tile * BN;
  if (row0 >= R) return;
  // pooling width of this tile: fixed (dense rows) or the class of the compact region the tile lies in
  int pool = g.pool;
  if (pool < 0)
    pool = row0 < g.hdr[1] ? 32 : row0 < g.hdr[2] ? 16 : row0 < g.hdr[3] ? 8 : row0 < g.hdr[4] ? 4 : row0 < g.hdr[5] ? 2 : 1;

  // ---- A loader: each thread owns rows (tid/4) [and (tid/4 + 64)], k-quad (tid%4) of the tile ----
  const int ar = tid >> 2, akq = tid & 3;
  const float *arow[NA];
  float csub[NA][3];
  // gathered rows [x - cx, y - cy, z - cz, f..]: the chain takes the feature columns first and the three relative
  // coordinates LAST (the oracle's chain_k): slab 0 carries zeros in their place, a closing 4-deep slab adds them
  const bool rot = g.mode != DET6D_A_ROWS && g.k > 3;
  float rel[NA][3];
#pragma unroll
  for (int i = 0; i < NA; ++i) rel[i][0] = rel[i][1] = rel[i][2] = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = row0 + ar + 64 * i;
    arow[i] = nullptr;
    csub[i][0] = csub[i][1] = csub[i][2] = 0.f;
    if (r < R) {
      if (g.mode == DET6D_A_GROUPED) {
        // ns and m are powers of two in every Det6D layer: shifts instead of two ~20-instruction integer
        // divisions per row (uniform branches; vector-ALU work is matrix time here)
        int cj, bi;
        if ((g.ns & (g.ns - 1)) == 0) cj = r >> __builtin_ctz(g.ns); else cj = r / g.ns;
        if ((g.m & (g.m - 1)) == 0) bi = cj >> __builtin_ctz(g.m); else bi = cj / g.m;
        const int p = g.idx[r];
        arow[i] = g.a + ((size_t)bi * g.n + p) * g.lda;
        if (akq == 0) {
          const float *c = g.ctr + (size_t)cj * g.ldctr;
          csub[i][0] = c[0]; csub[i][1] = c[1]; csub[i][2] = c[2];
        }
      } else if (g.mode == DET6D_A_COMPACT) {
        arow[i] = g.a + (size_t)g.crow_p[r] * g.lda;
        const int cj = g.crow_c[r];
        if (akq == 0 && cj >= 0) {
          const float *c = g.ctr + (size_t)(cj & 0x1fffffff) * g.ldctr;
          csub[i][0] = c[0]; csub[i][1] = c[1]; csub[i][2] = c[2];
        }
      } else {
        arow[i] = g.a + (size_t)r * g.lda;
      }
    }
  }
  // ---- B loader: BK x BN tile = 4*BN float4 ----
  constexpr int NB4 = (BK * BN / 4 + 255) / 256;  // float4 per thread
  constexpr int B4_PER_ROW = BN / 4;

  float4 ra[NA][KU];
  float4 rb[NB4];

  // ---- fast loader: byte offsets relative to g.a / g.w, valid for every full slab ----
  uint32_t voff_a[NA][KU], voff_b[NB4];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int u = 0; u < KU; ++u)
      voff_a[i][u] = arow[i] ? (uint32_t)((arow[i] - g.a) + 4 * akq + 16 * u) * 4u : 0u;
#pragma unroll
  for (int i = 0; i < NB4; ++i) {
    const int f = tid + 256 * i;
    const int c = colb + 4 * (f % B4_PER_ROW);
    const bool ok = f < BK * B4_PER_ROW && c + 3 < g.ldw;
    voff_b[i] = ok ? (uint32_t)((f / B4_PER_ROW) * g.ldw + c) * 4u : 0u;
  }
  const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void *)g.a, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void *)g.w, 0, 0xffffffff, 0x00020000);
  auto load_tile_fast = [&](int k0) {   // requires k0 > 0 and k0 + BK <= K
    const int soff_a = k0 * 4, soff_b = k0 * g.ldw * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(srd_a, voff_a[i][u], soff_a, 0));
        ra[i][u] = make_float4(v.x, v.y, v.z, v.w);
      }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(srd_w, voff_b[i], soff_b, 0));
      rb[i] = make_float4(v.x, v.y, v.z, v.w);
    }
  };

  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int k = k0 + 4 * akq + 16 * u;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (arow[i] != nullptr && k + 3 < K) v = *reinterpret_cast<const float4 *>(arow[i] + k);
        else if (arow[i] != nullptr && k < K) {  // K % 4 != 0 tail
          v.x = arow[i][k];
          if (k + 1 < K) v.y = arow[i][k + 1];
          if (k + 2 < K) v.z = arow[i][k + 2];
        }
        if (k0 == 0 && akq == 0 && u == 0) {  // grouped_xyz -= new_xyz (pointnet2_utils.py:449-450)
          v.x = v.x - csub[i][0]; v.y = v.y - csub[i][1]; v.z = v.z - csub[i][2];
          if (rot) {   // the relative coordinates enter the chain LAST (oracle: chain_k): kept for the closing mini-slab
            rel[i][0] = v.x; rel[i][1] = v.y; rel[i][2] = v.z;
            v.x = v.y = v.z = 0.f;
          }
        }
        ra[i][u] = v;
      }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int f = tid + 256 * i;
      const int k = k0 + f / B4_PER_ROW;
      const int c = colb + 4 * (f % B4_PER_ROW);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < BK * B4_PER_ROW && k < K && c + 3 < g.ldw) v = *reinterpret_cast<const float4 *>(g.w + (size_t)k * g.ldw + c);
      rb[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    float *As = As_all + buf * BK * LDA_S;
    float *Bs = Bs_all + buf * BK * BN;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        // transposed store As[k][row]: bank = (8*akq + 2*j + row) % 32 -> 32 distinct banks per half-wave
        float *dst = As + (4 * akq + 16 * u) * LDA_S + ar + 64 * i;
        dst[0 * LDA_S] = ra[i][u].x;
        dst[1 * LDA_S] = ra[i][u].y;
        dst[2 * LDA_S] = ra[i][u].z;
        dst[3 * LDA_S] = ra[i][u].w;
      }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int f = tid + 256 * i;
      if (f < BK * B4_PER_ROW)
        *reinterpret_cast<float4 *>(Bs + (f / B4_PER_ROW) * BN + 4 * (f % B4_PER_ROW)) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int kh = lane >> 5, l31 = lane & 31;
  const int arow_s = wm * 32 * TM + l31;
  const int bcol_s = wn * 32 * TN + l31;

  auto kstep = [&](const float *As, const float *Bs, int ks) {
    float af[TM], bf[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = As[(2 * ks + kh) * LDA_S + arow_s + 32 * i];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = Bs[(2 * ks + kh) * BN + bcol_s + 32 * j];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
  };
  auto compute = [&](int buf) {
    const float *As = As_all + buf * BK * LDA_S;
    const float *Bs = Bs_all + buf * BK * BN;
    if constexpr (PIPE) {
      float af[2][TM], bf[2][TN];
      auto frag = [&](int ks, float (&a_)[TM], float (&b_)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a_[i] = As[(2 * ks + kh) * LDA_S + arow_s + 32 * i];
#pragma unroll
        for (int j = 0; j < TN; ++j) b_[j] = Bs[(2 * ks + kh) * BN + bcol_s + 32 * j];
      };
      frag(0, af[0], bf[0]);
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        if (ks + 1 < BK / 2) frag(ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
      }
      // issue order: the LDS reads of k-step s + 1 go out BEFORE the MFMAs of k-step s (left alone the scheduler puts them
      // behind, into the same registers, and every k-step waits out the LDS latency)
      constexpr int NRD = (TM + 1) / 2 + (TN + 1) / 2;      // ds_read2_b32 pairs the fragments' two 32-row blocks
      __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        if (ks + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) kstep(As, Bs, ks);
    }
  };
  // A short last slab (K = 68, 132, 260: [xyz | C | pad] rows) is peeled out of the main loop and issues
  // only the k-steps that hold data; the skipped steps would multiply zeros, so every output's chain
  // is unchanged.  (A branch INSIDE the main loop costs 10-20 %: it breaks the load/MFMA overlap.)
  auto compute_tail = [&](int buf, int kleft) {
    const int nks = (kleft + 1) >> 1;
#pragma unroll 1
    for (int ks = 0; ks < nks; ++ks) kstep(As_all + buf * BK * LDA_S, Bs_all + buf * BK * BN, ks);
  };

  // empty-ball counts of the groups this wave pools (pooled layers): fetched now, consumed in the epilogue
  int pre_cnt[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) pre_cnt[i][qq] = 1;
  if (FAST_EPI && g.pool > 0 && g.cnt && row0 + BM <= R) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rbase = __builtin_amdgcn_readfirstlane(row0 + (tid >> 6) / WNW * 32 * TM + 32 * i);   // wave-uniform
      const int ngrp = 32 / g.pool;
      for (int qq = 0; qq < 4; ++qq)
        if (qq < ngrp) pre_cnt[i][qq] = g.cnt[rbase / g.pool + qq];
    }
  }
  // compact rows: the centre each of this lane's (up to four) pooled values belongs to, -1 = nothing to store
  // (alignment rows, or another lane is the writer).  Rows of a 32x32 tile held by a lane: 8*qq + 4*kh + (0..3).
  //   class 4: group = rows 8*qq + 4*kh .. +3, every lane writes its own four groups
  //   class 8 / 16 / 32: the lane^32 exchange completes the group; lanes of half 0 write group qq / qq>>1 / 0
  int pre_ctr[TM][4];
  if (g.pool < 0 && pool >= 4) {
    const int khl = (tid & 63) >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rbase = row0 + wm * 32 * TM + 32 * i;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int rsel = pool == 4 ? 8 * qq + 4 * khl : pool == 8 ? 8 * qq : pool == 16 ? 16 * (qq >> 1) : 0;
        const bool writer = pool == 4 || (khl == 0 && (pool == 8 || (pool == 16 && !(qq & 1)) || (pool == 32 && qq == 0)));
        int cj = -1;
        if (writer) cj = g.crow_c[rbase + rsel];
        pre_ctr[i][qq] = cj < 0 ? -1 : (cj & 0x3fffffff);
        pre_cnt[i][qq] = (cj & 0x40000000) ? 0 : 1;   // bit 30: empty ball
      }
    }
  }
  if (FAST && BK <= K) {
    // slab 0 through the fast loader too; grouped_xyz -= new_xyz touches the first float4 of each row only
    const int soff0 = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(srd_a, voff_a[i][u], soff0, 0));
        ra[i][u] = make_float4(v.x, v.y, v.z, v.w);
      }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(srd_w, voff_b[i], soff0, 0));
      rb[i] = make_float4(v.x, v.y, v.z, v.w);
    }
    if (g.mode != DET6D_A_ROWS && akq == 0) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        ra[i][0].x = ra[i][0].x - csub[i][0]; ra[i][0].y = ra[i][0].y - csub[i][1]; ra[i][0].z = ra[i][0].z - csub[i][2];
        if (rot) {
          rel[i][0] = ra[i][0].x; rel[i][1] = ra[i][0].y; rel[i][2] = ra[i][0].z;
          ra[i][0].x = ra[i][0].y = ra[i][0].z = 0.f;
        }
      }
    }
  } else {
    load_tile(0);
  }
  if (NBUF == 1) {
    int k0 = 0;
    if (FAST) {   // steady state: the next slab is full, fetched by the predicate-free loader
      for (; k0 + 2 * BK <= K; k0 += BK) {
        store_tile(0);
        __syncthreads();
        load_tile_fast(k0 + BK);
        if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);     // the loads stay above the MFMA block
        compute(0);
        __syncthreads();
      }
    }
    for (; k0 + BK <= K; k0 += BK) {   // FAST: only the last full slab (its successor, if any, is the short one)
      store_tile(0);
      __syncthreads();
      if (k0 + BK < K) load_tile(k0 + BK);
      compute(0);
      __syncthreads();
    }
    if (k0 < K) {
      store_tile(0);
      __syncthreads();
      compute_tail(0, K - k0);
    }
  } else {
    // two LDS buffers: ONE barrier per slab (the slab after next is written while nobody reads it)
    store_tile(0);
    __syncthreads();
    int buf = 0, k0 = 0;
    if (FAST) {
      for (; k0 + 2 * BK <= K; k0 += BK) {
        load_tile_fast(k0 + BK);
        if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
        compute(buf);
        store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
      }
    }
    for (; k0 + BK <= K; k0 += BK) {
      const bool more = k0 + BK < K;
      if (more) load_tile(k0 + BK);
      compute(buf);
      if (more) store_tile(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    if (k0 < K) compute_tail(buf, K - k0);
  }

  if (rot) {   // closing slab: k = 0, 1, 2 of the gathered rows (relative coordinates) x weight rows 0..2
    __syncthreads();
    if (akq == 0) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        float *dst = As_all + ar + 64 * i;
        dst[0 * LDA_S] = rel[i][0]; dst[1 * LDA_S] = rel[i][1]; dst[2 * LDA_S] = rel[i][2]; dst[3 * LDA_S] = 0.f;
      }
    }
    if (tid < BN) {
      const int c = colb + tid;
      const bool ok = c < g.ldw;
      Bs_all[0 * BN + tid] = ok ? g.w[c] : 0.f;
      Bs_all[1 * BN + tid] = ok ? g.w[(size_t)g.ldw + c] : 0.f;
      Bs_all[2 * BN + tid] = ok ? g.w[(size_t)2 * g.ldw + c] : 0.f;
      Bs_all[3 * BN + tid] = 0.f;
    }
    __syncthreads();
    kstep(As_all, Bs_all, 0);
    kstep(As_all, Bs_all, 1);
  }

  // ---- epilogue ----
  // interior tiles of plain (non-pooled) layers: no bound predicates, no 64-bit address arithmetic — one
  // per-lane byte offset per 32x32 tile, the row of each accumulator register as a scalar offset
  if (FAST_EPI && g.pool == 0 && row0 + BM <= R && colb + BN <= N) {
    const __amdgpu_buffer_rsrc_t srd_y = __builtin_amdgcn_make_buffer_rsrc((void *)g.y, 0, 0xffffffff, 0x00020000);
    const int ldy4 = g.ldy * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = colb + wn * 32 * TN + 32 * j + l31;
      const float sh = g.shift ? g.shift[col] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int rbase = row0 + wm * 32 * TM + 32 * i;
        const uint32_t voff = (uint32_t)((rbase + 4 * kh) * g.ldy + g.col0 + col) * 4u;
        if (g.act == 1) {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d6_relu(acc[i][j][e] + sh)), srd_y, voff,
                                                  ((e & 3) + 8 * (e >> 2)) * ldy4, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[i][j][e] + sh), srd_y, voff,
                                                  ((e & 3) + 8 * (e >> 2)) * ldy4, 0);
        }
      }
    }
    return;
  }
  // interior tiles of pooled layers (the last layer of every SA group, 45 % of the GEMM time): the empty-ball
  // counts were fetched before the K loop, the lane^32 exchange is one v_permlane32_swap + one v_max, stores go
  // through the buffer path with the group row as a scalar offset, no bound predicates
  if (FAST_EPI && g.pool > 0 && row0 + BM <= R && colb + BN <= N) {
    const __amdgpu_buffer_rsrc_t srd_y = __builtin_amdgcn_make_buffer_rsrc((void *)g.y, 0, 0xffffffff, 0x00020000);
    const int ldy4 = g.ldy * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = colb + wn * 32 * TN + 32 * j + l31;
      const float sh = g.shift ? g.shift[col] : 0.f;
      const uint32_t voff = (uint32_t)(g.col0 + col) * 4u;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float q[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          float lo = d6_vmax(d6_vmax(acc[i][j][4 * qq], acc[i][j][4 * qq + 1]), d6_vmax(acc[i][j][4 * qq + 2], acc[i][j][4 * qq + 3]));
          float hi = lo;
          // lo' = [lo.lanes0-31 | hi.lanes0-31], hi' = [lo.lanes32-63 | hi.lanes32-63]: max(lo', hi') is the 8-row maximum in BOTH halves
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
          q[qq] = d6_vmax(lo, hi);
        }
        const int rbase = row0 + wm * 32 * TM + 32 * i;
        if (g.pool == 32) {
          const float m = relu_act(d6_vmax(d6_vmax(q[0], q[1]), d6_vmax(q[2], q[3])) + sh, g.act);
          const float v = pre_cnt[i][0] > 0 ? m : 0.f;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), srd_y, voff, (rbase / 32) * ldy4, 0);
        } else if (g.pool == 16) {
          const float m0 = relu_act(d6_vmax(q[0], q[1]) + sh, g.act), m1 = relu_act(d6_vmax(q[2], q[3]) + sh, g.act);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pre_cnt[i][0] > 0 ? m0 : 0.f), srd_y, voff,
                                                (rbase / 16) * ldy4, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pre_cnt[i][1] > 0 ? m1 : 0.f), srd_y, voff,
                                                (rbase / 16 + 1) * ldy4, 0);
        } else {
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const float m = relu_act(q[qq] + sh, g.act);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pre_cnt[i][qq] > 0 ? m : 0.f), srd_y, voff,
                                                  (rbase / 8 + qq) * ldy4, 0);
          }
        }
      }
    }
    return;
  }
  // compact rows, classes 1 and 2 (single hits / pairs: the cheap narrow groups): every accumulator (pair) is a
  // group of its own; the row tags are fetched here
  if (g.pool < 0 && pool < 4) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = colb + wn * 32 * TN + 32 * j + l31;
      const bool cok = col < N;
      const float sh = (cok && g.shift) ? g.shift[col] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int rbase = row0 + wm * 32 * TM + 32 * i;
#pragma unroll
        for (int e = 0; e < 16; e += 1) {
          if (pool == 2 && (e & 1)) continue;
          const float raw = pool == 2 ? d6_vmax(acc[i][j][e], acc[i][j][e + 1]) : acc[i][j][e];
          const int cj = g.crow_c[rbase + (e & 3) + 8 * (e >> 2) + 4 * kh];
          if (cok && cj >= 0) {
            const float val = (cj & 0x40000000) ? 0.f : relu_act(raw + sh, g.act);
            float *dst = g.y + (size_t)(cj & 0x1fffffff) * g.ldy + g.col0 + col;
            if (cj & 0x20000000) __hip_atomic_fetch_max(reinterpret_cast<int *>(dst), __builtin_bit_cast(int, val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = val;
          }
        }
      }
    }
    return;
  }
  // compact rows, pooled layer: segment maxima by class, scattered to the rows of their centres
  if (g.pool < 0) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = colb + wn * 32 * TN + 32 * j + l31;
      const bool cok = col < N;
      const float sh = (cok && g.shift) ? g.shift[col] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float v[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          v[qq] = d6_vmax(d6_vmax(acc[i][j][4 * qq], acc[i][j][4 * qq + 1]), d6_vmax(acc[i][j][4 * qq + 2], acc[i][j][4 * qq + 3]));
        if (pool > 4) {
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            float lo = v[qq], hi = v[qq];
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
            v[qq] = d6_vmax(lo, hi);
          }
          if (pool == 16) {
            v[0] = d6_vmax(v[0], v[1]);
            v[2] = d6_vmax(v[2], v[3]);
          } else if (pool == 32) {
            v[0] = d6_vmax(d6_vmax(v[0], v[1]), d6_vmax(v[2], v[3]));
          }
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int cj = pre_ctr[i][qq];
          if (cok && cj >= 0) {
            const float val = pre_cnt[i][qq] > 0 ? relu_act(v[qq] + sh, g.act) : 0.f;
            float *dst = g.y + (size_t)(cj & 0x1fffffff) * g.ldy + g.col0 + col;
            // bit 29: the centre's rows are cut into several parts (compact.hip, split lists): maximum over the parts
            // by an integer atomic max on the non-negative post-ReLU values (the buffer was zeroed)
            if (cj & 0x20000000) __hip_atomic_fetch_max(reinterpret_cast<int *>(dst), __builtin_bit_cast(int, val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = val;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = colb + wn * 32 * TN + 32 * j + l31;
    const bool cok = col < N;
    const float sh = (cok && g.shift) ? g.shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rbase = row0 + wm * 32 * TM + 32 * i;  // first row of this 32x32 tile
      if (g.pool == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = rbase + (e & 3) + 8 * (e >> 2) + 4 * kh;
          if (cok && row < R) g.y[(size_t)row * g.ldy + g.col0 + col] = relu_act(acc[i][j][e] + sh, g.act);
          else if (col < g.ncols_pad && row < R) g.y[(size_t)row * g.ldy + g.col0 + col] = 0.f;   // padding columns of the next layer's K
        }
      } else {
        // rows of a tile: (e&3) + 8*(e>>2) + 4*kh  -> 8-row bundle q = e>>2 spans both lane halves.
        // x -> act(x + shift) is monotone, so the max over the rows is taken on the raw accumulators and
        // shift / ReLU are applied to the pooled value only (identical result, a third of the vector ops).
        float q[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          float m = d6_vmax(d6_vmax(acc[i][j][4 * qq], acc[i][j][4 * qq + 1]), d6_vmax(acc[i][j][4 * qq + 2], acc[i][j][4 * qq + 3]));
          q[qq] = d6_vmax(m, __shfl_xor(m, 32));
        }
        if (g.pool == 32) {
          const float m = relu_act(d6_vmax(d6_vmax(q[0], q[1]), d6_vmax(q[2], q[3])) + sh, g.act);
          const int grp = rbase / 32;
          if (cok && kh == 0 && rbase < R) {
            const bool live = g.cnt ? g.cnt[grp] > 0 : true;
            g.y[(size_t)grp * g.ldy + g.col0 + col] = live ? m : 0.f;
          }
        } else if (g.pool == 16) {
          const float m0 = relu_act(d6_vmax(q[0], q[1]) + sh, g.act);
          const float m1 = relu_act(d6_vmax(q[2], q[3]) + sh, g.act);
          const int grp = rbase / 16;
          if (cok && kh == 0) {
            if (rbase < R) {
              const bool live = g.cnt ? g.cnt[grp] > 0 : true;
              g.y[(size_t)grp * g.ldy + g.col0 + col] = live ? m0 : 0.f;
            }
            if (rbase + 16 < R) {
              const bool live = g.cnt ? g.cnt[grp + 1] > 0 : true;
              g.y[(size_t)(grp + 1) * g.ldy + g.col0 + col] = live ? m1 : 0.f;
            }
          }
        } else {  // pool == 8
          const int grp = rbase / 8;
          if (cok && kh == 0) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
              if (rbase + 8 * qq < R) {
                const bool live = g.cnt ? g.cnt[grp + qq] > 0 : true;
                g.y[(size_t)(grp + qq) * g.ldy + g.col0 + col] = live ? relu_act(q[qq] + sh, g.act) : 0.f;
              }
          }
        }
      }
    }
  }
}

}  // namespace

DET6D_API int det6d_linear(const det6d_linear_args *a, det6d_stream_t stream) {
  D6_GEMM_PRIO_HOST();
  if (!a || a->rows < 0 || a->k <= 0 || a->ncols <= 0 || !a->a || !a->w || !a->y) return DET6D_EINVAL;
  if ((a->lda & 3) || (a->ldw & 3) || ((uintptr_t)a->a & 15) || ((uintptr_t)a->w & 15)) return DET6D_EINVAL;
  if (a->k > a->lda || a->ncols > a->ldw) return DET6D_EINVAL;
  if (a->ncols_pad && (a->ncols_pad < a->ncols || a->ncols_pad > a->ncols + 3 || a->col0 + a->ncols_pad > a->ldy || a->pool)) return DET6D_EINVAL;
  if (a->mode == DET6D_A_GROUPED) {
    if (!a->idx || !a->ctr || a->ns <= 0 || a->m <= 0 || a->n <= 0 || a->ldctr < 3) return DET6D_EINVAL;
    if (a->rows % (a->m * a->ns)) return DET6D_EINVAL;
  } else if (a->mode == DET6D_A_COMPACT) {
    if (!a->hdr || !a->crow_p || !a->crow_c || !a->ctr || a->n <= 0 || a->ldctr < 3) return DET6D_EINVAL;
  } else if (a->mode != DET6D_A_ROWS) {
    return DET6D_EINVAL;
  }
  if (a->pool < 0) {   // class pooling over compact rows (parts combined by an integer max: ReLU outputs only)
    if (!a->hdr || !a->crow_c || a->act != 1) return DET6D_EINVAL;
  } else {
    if (a->pool != 0 && a->pool != 8 && a->pool != 16 && a->pool != 32) return DET6D_EINVAL;
    if (a->pool && (a->rows % a->pool)) return DET6D_EINVAL;
  }
  if (a->hdr && (a->rows & 127)) return DET6D_EINVAL;   // capacity of a compact row space (det6d_compact_rows_capacity)
  if (a->rows == 0) return DET6D_OK;
  hipStream_t s = (hipStream_t)stream;
  // tile choice by the EXPECTED live rows: a compact list is sized for the dense row space (capacity) but holds a
  // fraction of it (DET6D_COMPACT_ROWS_EST = expected capacity / live ratio; the grid still covers the capacity)
  static const int rows_est_div = det6d_env_int("DET6D_COMPACT_ROWS_EST", 1);
  const int rows_est = a->hdr && rows_est_div > 1 ? a->rows / rows_est_div : a->rows;
  const int gm = det6d_divup(rows_est, 128);
  const int gm_cap = det6d_divup(a->rows, 128);
  // the buffer-load fast path addresses A and W with 32-bit byte offsets
  const size_t a_rows = a->mode == DET6D_A_GROUPED ? (size_t)(a->rows / (a->m * a->ns)) * a->n
                        : a->mode == DET6D_A_COMPACT ? (size_t)a->n : (size_t)a->rows;
  const bool fits32 = a_rows * a->lda * 4 < 0xfff00000ull && (size_t)a->k * a->ldw * 4 < 0xfff00000ull &&
                      (size_t)a->rows * a->ldy * 4 < 0xfff00000ull;
  static const bool no_fast = det6d_env_set("DET6D_LINEAR_NO_FAST");
  if (!fits32 || no_fast) {   // same tiles, plain predicated loader
    if (a->ncols > 64) {
      if (gm * det6d_divup(a->ncols, 128) < 256)
        hipLaunchKernelGGL((linear_kernel<64, 64, 2, 2, 1, 1, 16, 1, 0>), dim3(det6d_divup(a->rows, 64) * det6d_divup(a->ncols, 64)),
                           dim3(256), 0, s, *a);
      else if (a->ncols <= 512 && a->k <= 256)
        hipLaunchKernelGGL((linear_kernel<128, 64, 2, 2, 2, 1, 16, 1, 0>), dim3(gm_cap * det6d_divup(a->ncols, 64)), dim3(256), 0, s, *a);
      else
        hipLaunchKernelGGL((linear_kernel<128, 128, 2, 2, 2, 2, 16, 1, 0>), dim3(gm_cap * det6d_divup(a->ncols, 128)), dim3(256), 0, s, *a);
    } else if (a->ncols > 32) {
      if (gm < 128)
        hipLaunchKernelGGL((linear_kernel<64, 64, 2, 2, 1, 1, 16, 1, 0>), dim3(det6d_divup(a->rows, 64)), dim3(256), 0, s, *a);
      else
        hipLaunchKernelGGL((linear_kernel<128, 64, 2, 2, 2, 1, 16, 1, 0>), dim3(gm_cap), dim3(256), 0, s, *a);
    } else {
      hipLaunchKernelGGL((linear_kernel<128, 32, 4, 1, 1, 1, 16, 1, 0>), dim3(gm_cap), dim3(256), 0, s, *a);
    }
    return det6d_check_launch("det6d_linear");
  }
  // compact lists (a->hdr): launches of 7-43 k live rows; 128x64 tiles fill the idle chip better when such a launch runs
  // alone (GEMM family 52 -> 56 TF stand-alone) and are neutral with 16 passes in flight (9930 vs 9920 scenes/s, family
  // at saturation 90.6 vs 90.2 TF), so they are the default there; dense rows keep 128x128
  static const int k64_env = det6d_env_int("DET6D_LINEAR_K64MAX", -1);
  static const int n64_env = det6d_env_int("DET6D_LINEAR_N64MAX", -1);
  const int force_k_max = k64_env >= 0 ? k64_env : (a->hdr ? 512 : 0);
  static const bool fast64 = (det6d_env_int("DET6D_LINEAR_FAST64", 0) != 0);
  static const int pipe = det6d_env_int("DET6D_LINEAR_PIPE", 1);            // round 6: software-pipelined slab (FASTLVL 3): 596 -> 580 us over the four plain GEMMs of an 80-scene pass
  static const bool nbuf2 = det6d_env_set("DET6D_LINEAR_NBUF2");   // double-buffered LDS tiles, one barrier per slab
  static const int bk32 = det6d_env_int("DET6D_LINEAR_BK32", 0);   // K from which BK = 32 is used
  const int force_n_max = n64_env >= 0 ? n64_env : (a->hdr ? 1024 : 512);
  // Round 6: a grid of 128 x 128 tiles that leaves the CUs with 2.5 workgroups each (the head's shared FC over the 20480 rows of
  // an 80-scene pass: 160 x 4 = 640 workgroups on 256 CUs) runs three rounds where two and a half would do; 128 x 64 tiles make
  // it 5 per CU: 270.9 -> 242.8 us per launch replayed back to back (119 -> 133 TFLOP/s, scripts/r06/gpu_t8.sh).  Only long K
  // loops gain (SA3's aggregation FC, 640 workgroups as well but K = 512, does not), so the rule asks for K >= 1024.
  static const int balance_env = det6d_env_int("DET6D_LINEAR_BALANCE64", 1);
  const int wg128 = gm_cap * det6d_divup(a->ncols, 128);
  const bool balance64 = balance_env && !a->hdr && a->k >= 1024 && a->ncols >= 128 && (a->ncols & 127) == 0 && wg128 >= 256 && wg128 < 1024 &&
                         (wg128 % 256) != 0 && ((2 * wg128) % 256) == 0;
  if (a->ncols > 64) {
    // few row tiles (the FC layers over 256..1024 centres per scene): 64x64 tiles spread the K loop
    // over all CUs instead of leaving most of the chip idle behind a handful of 128x128 tiles.  These
    // launches are latency-bound (one wave per SIMD): the buffer-load fast path measured 5-9 % SLOWER
    // there, so they keep the plain loader (FAST = false).
    if (gm * det6d_divup(a->ncols, 128) < 256) {
      if (fast64)
        hipLaunchKernelGGL((linear_kernel<64, 64, 2, 2, 1, 1, 16, 1, 2>), dim3(det6d_divup(a->rows, 64) * det6d_divup(a->ncols, 64)),
                           dim3(256), 0, s, *a);
      else
        hipLaunchKernelGGL((linear_kernel<64, 64, 2, 2, 1, 1, 16, 1, 1>), dim3(det6d_divup(a->rows, 64) * det6d_divup(a->ncols, 64)),
                           dim3(256), 0, s, *a);
    }
    else if ((a->ncols <= force_n_max && a->k <= force_k_max) || balance64) {
      // 128x64 tiles (5 waves/SIMD) used to win 2-7 % on short K loops; with the vector-ALU-free main loop
      // and epilogue the 128x128 tile is ahead everywhere (GEMM family 1.921 -> 1.906 ms), so this branch is
      // off by default (DET6D_LINEAR_K64MAX = largest K that still takes it)
      if (pipe) hipLaunchKernelGGL((linear_kernel<128, 64, 2, 2, 2, 1, 16, 1, 3>), dim3(gm_cap * det6d_divup(a->ncols, 64)), dim3(256), 0, s, *a);
      else hipLaunchKernelGGL((linear_kernel<128, 64, 2, 2, 2, 1>), dim3(gm_cap * det6d_divup(a->ncols, 64)), dim3(256), 0, s, *a);
    } else if (nbuf2) {
      if (pipe) hipLaunchKernelGGL((linear_kernel<128, 128, 2, 2, 2, 2, 16, 2, 3>), dim3(gm_cap * det6d_divup(a->ncols, 128)), dim3(256), 0, s, *a);
      else hipLaunchKernelGGL((linear_kernel<128, 128, 2, 2, 2, 2, 16, 2>), dim3(gm_cap * det6d_divup(a->ncols, 128)), dim3(256), 0, s, *a);
    } else if (bk32 && a->k >= bk32) {
      if (pipe) hipLaunchKernelGGL((linear_kernel<128, 128, 2, 2, 2, 2, 32, 1, 3>), dim3(gm_cap * det6d_divup(a->ncols, 128)), dim3(256), 0, s, *a);
      else hipLaunchKernelGGL((linear_kernel<128, 128, 2, 2, 2, 2, 32>), dim3(gm_cap * det6d_divup(a->ncols, 128)), dim3(256), 0, s, *a);
    } else if (pipe)
      hipLaunchKernelGGL((linear_kernel<128, 128, 2, 2, 2, 2, 16, 1, 3>), dim3(gm_cap * det6d_divup(a->ncols, 128)), dim3(256), 0, s, *a);
    else
      hipLaunchKernelGGL((linear_kernel<128, 128, 2, 2, 2, 2>), dim3(gm_cap * det6d_divup(a->ncols, 128)), dim3(256), 0, s, *a);
  } else if (a->ncols > 32) {
    if (gm < 128 && fast64)
      hipLaunchKernelGGL((linear_kernel<64, 64, 2, 2, 1, 1, 16, 1, 2>), dim3(det6d_divup(a->rows, 64)), dim3(256), 0, s, *a);
    else if (gm < 128)
      hipLaunchKernelGGL((linear_kernel<64, 64, 2, 2, 1, 1, 16, 1, 1>), dim3(det6d_divup(a->rows, 64)), dim3(256), 0, s, *a);
    else
      hipLaunchKernelGGL((linear_kernel<128, 64, 2, 2, 2, 1>), dim3(gm_cap), dim3(256), 0, s, *a);
  } else {
    hipLaunchKernelGGL((linear_kernel<128, 32, 4, 1, 1, 1>), dim3(gm_cap), dim3(256), 0, s, *a);
  }
  return det6d_check_launch("det6d_linear");
}
