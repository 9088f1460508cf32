// fps.hip — farthest point sampling for gfx950 (wave64, DPP reductions, one barrier per round).
//
// Replaces farthest_point_sampling_kernel / furthest_point_sampling_weights_kernel of
// core/pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:101-222,419-540.
//
// Design (MI355X-first, not a translation of the CUDA kernel):
//  * one workgroup per scene; every point lives in VGPRs for the whole kernel (x,y,z,min-dist,
//    up to 16 points per lane at N=16384), so a round touches no memory except two 16-byte LDS
//    slots per wave;
//  * the reference resolves ties through the order of its strided scan + shared-memory halving
//    tree: among equal maxima the winner minimises (bitrev(k mod S), k), S = block size.  We get
//    exactly that order for free by giving hardware thread h the reference's virtual thread
//    v = bitrev(h): then "lowest hardware lane / lowest wave wins" is the reference's rule, and the
//    argmax becomes  wave max (6 DPP steps) -> ballot -> s_ff1 -> v_readlane;
//  * the winner's coordinates travel with the (value, index) pair through LDS, so the next round
//    starts without a dependent global load;
//  * LDS slots are double-buffered by round parity: one s_barrier per round.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kMaxPPT = 16;

__device__ __forceinline__ unsigned bitrev_n(unsigned v, int bits) {
  return bits == 0 ? 0u : (__builtin_bitreverse32(v) >> (32 - bits));
}

// Strided batch view + fused pre/post steps (used by det6d_fps_fused; the reference-shaped entry
// points pass the dense defaults): per-scene strides in elements, an index offset added on output
// (sample ranges, pointnet2_modules.py:380,448-450), implicit 1e10 min-distance initialisation, and
// s-fps weights computed on the fly as sigmoid(score)**gamma (pointnet2_modules.py:415-419).
struct FpsView {
  long long xyz_bstride, w_bstride, temp_bstride, idx_bstride;
  int idx_add;
  int init_temp;
  int w_is_score;
  float gamma;
};

struct Slot {
  float val;
  int idx;
  float x, y, z;
  float pad[3];
};

// PPT points per thread in registers. WEIGHTED selects the S-FPS scoring rule.
template <int PPT, bool WEIGHTED>
__global__ __launch_bounds__(1024) void fps_reg_kernel(int n, int m, int log2s,
                                                       const float *__restrict__ xyz,
                                                       const float *__restrict__ weights,
                                                       float *__restrict__ temp,
                                                       int *__restrict__ idxs, const FpsView vw) {
  typedef float vecf __attribute__((ext_vector_type(PPT)));
  __shared__ Slot slots[2][16];

  const int S = 1 << log2s;
  const int h = threadIdx.x;
  const int lane = h & 63;
  const int wave = h >> 6;
  const int nwaves = (blockDim.x + 63) >> 6;
  const bool live = h < S;
  const int v = (int)bitrev_n((unsigned)h, log2s);

  xyz += (size_t)blockIdx.x * vw.xyz_bstride;
  if (temp) temp += (size_t)blockIdx.x * vw.temp_bstride;
  idxs += (size_t)blockIdx.x * vw.idx_bstride;
  if (WEIGHTED) weights += (size_t)blockIdx.x * vw.w_bstride;

  vecf px, py, pz, pt;
  double pw[PPT];
  float pwf[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = v + (j << log2s);
    const bool ok = live && k < n;
    px[j] = ok ? xyz[(size_t)k * 3 + 0] : 0.f;
    py[j] = ok ? xyz[(size_t)k * 3 + 1] : 0.f;
    pz[j] = ok ? xyz[(size_t)k * 3 + 2] : 0.f;
    pt[j] = ok ? (vw.init_temp ? 1e10f : temp[k]) : 0.f;
    if (WEIGHTED) {
      pwf[j] = ok ? weights[k] : 0.f;
      if (vw.w_is_score) pwf[j] = d6_sigmoid_powf(pwf[j], vw.gamma);
      // max(weights[k], 1e-12) with a double literal (sampling_gpu.cu:466); fmax drops a NaN weight
      pw[j] = fmax((double)pwf[j], 1e-12);
    }
  }

  float cx = 0.f, cy = 0.f, cz = 0.f;
  int first_round;
  if (WEIGHTED) {
    first_round = 0;
  } else {
    first_round = 1;
    if (h == 0) idxs[0] = vw.idx_add;
    cx = xyz[0]; cy = xyz[1]; cz = xyz[2];
  }

  for (int r = first_round; r < m; ++r) {
    float best = live ? -1.0f : -__builtin_inff();
    int bj = 0;  // which of my points is my best (index into the register arrays)
    if (WEIGHTED && r == 0) {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const int k = v + (j << log2s);
        const float d = pwf[j];
        const bool up = live && k < n && d > best;
        bj = up ? j : bj;
        best = up ? d : best;
      }
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const int k = v + (j << log2s);
        const float d = d6_sqdist(px[j] - cx, py[j] - cy, pz[j] - cz);
        const float t = d6_fminf(d, pt[j]);
        pt[j] = t;
        float score = t;
        if (WEIGHTED) score = (float)((double)t * pw[j]);
        const bool up = live && k < n && score > best;
        bj = up ? j : bj;
        best = up ? score : best;
      }
    }
    // a live thread that found nothing keeps (-1, k=0) like the reference's (best=-1, besti=0)
    const bool found = best > -1.0f;
    const int bk = found ? v + (bj << log2s) : 0;

    // ---- wave argmax: lowest lane among the maxima wins ----
    const float wmax = d6_wave_max(best);
    const unsigned long long tie = __ballot(best == wmax);
    const int wl = __builtin_ctzll(tie);
    const int wj = d6_readlane_i(found ? bj : -1, wl);
    const int wk = d6_readlane_i(bk, wl);
    float sx, sy, sz;
    if (wj >= 0) {
      sx = d6_readlane_f(px[wj], wl);
      sy = d6_readlane_f(py[wj], wl);
      sz = d6_readlane_f(pz[wj], wl);
    } else {  // nothing beat -1: the reference then picks point 0
      sx = xyz[0]; sy = xyz[1]; sz = xyz[2];
    }
    Slot *sl = slots[r & 1];
    if (lane == 0) {
      sl[wave].val = wmax;
      sl[wave].idx = wk;
      sl[wave].x = sx; sl[wave].y = sy; sl[wave].z = sz;
    }
    __syncthreads();
    // ---- cross-wave argmax: lowest wave among the maxima wins ----
    const int src = lane < nwaves ? lane : 0;
    const float v2 = lane < nwaves ? sl[src].val : -__builtin_inff();
    const int i2 = sl[src].idx;
    const float x2 = sl[src].x, y2 = sl[src].y, z2 = sl[src].z;
    const float bmax = d6_wave_max(v2);
    const unsigned long long tie2 = __ballot(v2 == bmax);
    const int ww = __builtin_ctzll(tie2);
    const int old = d6_readlane_i(i2, ww);
    cx = d6_readlane_f(x2, ww);
    cy = d6_readlane_f(y2, ww);
    cz = d6_readlane_f(z2, ww);
    if (h == 0) idxs[r] = old + vw.idx_add;
  }

  // leave the final min-distances in temp like the reference does
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = v + (j << log2s);
    if (live && k < n && temp) temp[k] = pt[j];
  }
}


typedef float f32x2 __attribute__((ext_vector_type(2)));

// coordinates of slot `ws` (wave-uniform) of lane `wl`: scalar branches down to one slot
template <int LO, int HI, int N>
__device__ __forceinline__ void pick_slot(int ws, int wl, const float (&px)[N], const float (&py)[N],
                                          const float (&pz)[N], float &sx, float &sy, float &sz) {
  if constexpr (HI - LO == 1) {
    sx = d6_readlane_f(px[LO], wl);
    sy = d6_readlane_f(py[LO], wl);
    sz = d6_readlane_f(pz[LO], wl);
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (ws < MID) pick_slot<LO, MID>(ws, wl, px, py, pz, sx, sy, sz);
    else pick_slot<MID, HI>(ws, wl, px, py, pz, sx, sy, sz);
  }
}

// Fat-thread fast path.  The reference's block has S = opt_n_threads(N) virtual threads; their
// tie priority is p = bitrev(v).  Here T = 2^LOG2T <= S hardware threads each emulate S/T virtual
// threads with CONSECUTIVE priorities p = h*S/T .. (h+1)*S/T - 1 and scan their SLOTS = N/T points in
// (p, j) order, so "lowest hardware thread, then first slot" is still exactly the reference's rule —
// but the per-round reduction/broadcast code (which every wave executes) runs on 8, 4 or 1 waves
// instead of 16, and a single-wave block (N <= 1024) needs no LDS and no barrier at all.
// Distances are evaluated two points at a time with packed fp32 ops (v_pk_add/mul/fma_f32).
//
// S-FPS score (sampling_gpu.cu:466): float(double(t) * max(double(w), 1e-12)).  While w >= 1e-12 the product of the two floats
// is EXACT in double (48 significand bits), so rounding it to float once IS the IEEE fp32 product t * w (fp32 denormals are
// kept in this build: .amdhsa_float_denorm_mode_32 3): FASTW scores a point with one v_mul_f32 instead of convert / fp64
// multiply / convert and holds the weights as 32-bit values (SLOTS fewer registers).  A scene that holds a weight below
// 1e-12 — or a NaN weight, which the reference's max() turns into 1e-12 — cannot take that form: the FASTW launch writes
// flags[scene] = 1 for it and returns, and the GUARDED exact-double launch behind it (fps_mem_kernel: it returns at once for
// every scene whose flag is 0) samples that scene — many times slower (min-distances and weights in memory), exact, and rare: a
// confidence logit below -27.6.  Why two kernels, and why the memory-resident one behind: with both scoring forms inlined in
// one kernel the allocator sizes it for their union (fat<9, 32>: 202 -> 256 registers + 464 bytes of scratch per lane; with
// the exact form as a rolled loop over memory inside the kernel the fp32 form lost its gain: 358 -> 393 us); and a guarded
// launch must be CHEAP TO PLACE — a launch that only looks at a flag still needs its workgroup's registers on a CU before it
// may start: the register-resident exact kernel (202 registers x 8 waves) waited 4.3 ms on average for them in the
// 65536-point pipeline, the memory-resident one holds ~20 registers per lane.
template <int LOG2T, int SLOTS, bool WEIGHTED, bool FASTW = false>
__global__ __launch_bounds__(1 << LOG2T) void fps_fat_kernel(int n, int m, int log2s, int log2pptv,
                                                             const float *__restrict__ xyz,
                                                             const float *__restrict__ weights,
                                                             float *__restrict__ temp,
                                                             int *__restrict__ idxs, const FpsView vw,
                                                             int *__restrict__ flags = nullptr) {
  static_assert(WEIGHTED || !FASTW, "FASTW is a form of the weighted sampler");
  static_assert(SLOTS >= 2 && SLOTS % 2 == 0, "pairs");
  constexpr int T = 1 << LOG2T;
  constexpr int NW = T / 64;
  constexpr int H = SLOTS / 2;
  __shared__ Slot slots[2][NW > 1 ? NW : 1];
  const int h = threadIdx.x;
  const int lane = h & 63;
  const int wave = h >> 6;
  const int log2vpt = log2s - LOG2T;            // virtual threads per hardware thread
  d6_sampler_priority();

  xyz += (size_t)blockIdx.x * vw.xyz_bstride;
  if (temp) temp += (size_t)blockIdx.x * vw.temp_bstride;
  idxs += (size_t)blockIdx.x * vw.idx_bstride;
  if (WEIGHTED) weights += (size_t)blockIdx.x * vw.w_bstride;

  auto slot_point = [&](int s) {  // slot -> point index k = v + S * j
    const int pl = s >> log2pptv, j = s & ((1 << log2pptv) - 1);
    const int v = (int)bitrev_n((unsigned)((h << log2vpt) + pl), log2s);
    return v + (j << log2s);
  };

  float px[SLOTS], py[SLOTS], pz[SLOTS];   // statically indexed only -> plain VGPRs
  float pt[SLOTS];
  double pw[FASTW ? 1 : SLOTS];
  float pwf[SLOTS];
  bool small = false;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int k = slot_point(s);
    px[s] = xyz[(size_t)k * 3 + 0];
    py[s] = xyz[(size_t)k * 3 + 1];
    pz[s] = xyz[(size_t)k * 3 + 2];
    // cut the values loose from the dwordx3 load tuple, otherwise the allocator keeps the triple
    // AND the (x_s, x_s+1) pair copies the packed ops need alive for the whole kernel
    asm volatile("" : "+v"(px[s]), "+v"(py[s]), "+v"(pz[s]));
    pt[s] = vw.init_temp ? 1e10f : temp[k];
    if (WEIGHTED) {
      pwf[s] = weights[k];
      if (vw.w_is_score) pwf[s] = d6_sigmoid_powf(pwf[s], vw.gamma);
      if (FASTW) small = small || !((double)pwf[s] >= 1e-12);
      else pw[s] = fmax((double)pwf[s], 1e-12);   // `max(weights[k], 1e-12)` in double, sampling_gpu.cu:466
    }
  }
  if (FASTW) {
    small = NW > 1 ? (__syncthreads_or(small ? 1 : 0) != 0) : (__ballot(small) != 0ull);
    if (h == 0) flags[(size_t)blockIdx.x * vw.temp_bstride] = small ? 1 : 0;
    if (small) return;                          // this scene is sampled by the exact-double launch behind this one
  }

  float cx = 0.f, cy = 0.f, cz = 0.f;
  int first_round = WEIGHTED ? 0 : 1;
  if (!WEIGHTED) {
    if (h == 0) idxs[0] = vw.idx_add;
    cx = xyz[0]; cy = xyz[1]; cz = xyz[2];
  }

  for (int r = first_round; r < m; ++r) {
    float best = -1.0f;
    int bs = 0;
    if (WEIGHTED && r == 0) {
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const bool up = pwf[s] > best;
        bs = up ? s : bs;
        best = up ? pwf[s] : best;
      }
    } else {
      const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
      for (int q = 0; q < H; ++q) {
        const f32x2 dx = f32x2{px[2 * q], px[2 * q + 1]} - c2x;
        const f32x2 dy = f32x2{py[2 * q], py[2 * q + 1]} - c2y;
        const f32x2 dz = f32x2{pz[2 * q], pz[2 * q + 1]} - c2z;
        f32x2 d = dy * dy;
        d = __builtin_elementwise_fma(dx, dx, d);
        d = __builtin_elementwise_fma(dz, dz, d);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int s = 2 * q + e;
          const float t = d6_vmin(d[e], pt[s]);
          pt[s] = t;
          float score = t;
          if (WEIGHTED) score = FASTW ? t * pwf[s] : (float)((double)t * pw[s]);
          const bool up = score > best;
          bs = up ? s : bs;
          best = up ? score : best;
        }
      }
    }
    // a thread that found nothing keeps (-1, k = 0) like the reference's (best = -1, besti = 0)
    const bool found = best > -1.0f;

    const float wmax = d6_wave_max(best);
    const unsigned long long tie = __ballot(best == wmax);
    const int wl = __builtin_ctzll(tie);
    const int ws = d6_readlane_i(found ? bs : -1, wl);
    int old;
    float sx, sy, sz;
    if (ws >= 0) {
      old = d6_readlane_i(slot_point(ws), wl);
      // ws is wave-uniform: a scalar binary search over the slots, then three v_readlane
      pick_slot<0, SLOTS>(ws, wl, px, py, pz, sx, sy, sz);
    } else {
      old = 0;
      sx = xyz[0]; sy = xyz[1]; sz = xyz[2];
    }
    if (NW > 1) {
      Slot *sl = slots[r & 1];
      if (lane == 0) {
        sl[wave].val = wmax;
        sl[wave].idx = old;
        sl[wave].x = sx; sl[wave].y = sy; sl[wave].z = sz;
      }
      __syncthreads();
      const int src = lane & (NW - 1);
      const float v2 = sl[src].val;
      const int i2 = sl[src].idx;
      const float x2 = sl[src].x, y2 = sl[src].y, z2 = sl[src].z;
      const float bmax = d6_row_max16(v2);    // NW <= 16: lanes 0..15 hold every wave's entry
      const unsigned long long tie2 = __ballot(v2 == bmax);
      const int ww = __builtin_ctzll(tie2);    // lowest wave among the maxima
      old = d6_readlane_i(i2, ww);
      sx = d6_readlane_f(x2, ww);
      sy = d6_readlane_f(y2, ww);
      sz = d6_readlane_f(z2, ww);
    }
    cx = sx; cy = sy; cz = sz;
    if (h == 0) idxs[r] = old + vw.idx_add;
  }
  // `temp` is scratch: the reference's callers discard it (pointnet2_utils.py:26-29), so the final
  // min-distances are not written back here (the generic kernels below still do).
}

// Generic fallback for N > 16 * 1024: same schedule, min-distances and coordinates stay in
// memory (L2) like the reference kernel.  Correct for any N; not a tuned path.
template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void fps_mem_kernel(int n, int m, int log2s,
                                                       const float *__restrict__ xyz,
                                                       const float *__restrict__ weights,
                                                       float *__restrict__ temp,
                                                       int *__restrict__ idxs, const FpsView vw,
                                                       const int guarded = 0) {
  __shared__ Slot slots[2][16];
  if (guarded) {      // behind an fp32-scoring launch (fps_fat_kernel<.., FASTW>): only the scenes it handed over (flag = the
                      // first word of the scene's scratch, which this kernel is about to initialise: read it first)
    const int flag = reinterpret_cast<const int *>(temp)[(size_t)blockIdx.x * vw.temp_bstride];
    if (flag == 0) return;
    __syncthreads();
  }
  const int S = 1 << log2s;
  const int h = threadIdx.x;
  const int lane = h & 63;
  const int wave = h >> 6;
  const int nwaves = (blockDim.x + 63) >> 6;
  const bool live = h < S;
  const int v = (int)bitrev_n((unsigned)h, log2s);
  xyz += (size_t)blockIdx.x * vw.xyz_bstride;
  if (temp) temp += (size_t)blockIdx.x * vw.temp_bstride;
  idxs += (size_t)blockIdx.x * vw.idx_bstride;
  if (WEIGHTED) weights += (size_t)blockIdx.x * vw.w_bstride;

  auto weight_of = [&](int k) {
    const float w = weights[k];
    return vw.w_is_score ? d6_sigmoid_powf(w, vw.gamma) : w;
  };
  if (vw.init_temp && live)
    for (int k = v; k < n; k += S) temp[k] = 1e10f;   // same thread re-reads its own entries: no sync needed
  int old = 0;
  int first_round = WEIGHTED ? 0 : 1;
  if (!WEIGHTED && h == 0) idxs[0] = vw.idx_add;
  for (int r = first_round; r < m; ++r) {
    const float cx = xyz[(size_t)old * 3 + 0], cy = xyz[(size_t)old * 3 + 1], cz = xyz[(size_t)old * 3 + 2];
    float best = live ? -1.0f : -__builtin_inff();
    int bk = 0;
    if (live) {
      for (int k = v; k < n; k += S) {
        float score;
        if (WEIGHTED && r == 0) {
          score = weight_of(k);
        } else {
          const float d = d6_sqdist(xyz[(size_t)k * 3 + 0] - cx, xyz[(size_t)k * 3 + 1] - cy,
                                    xyz[(size_t)k * 3 + 2] - cz);
          const float t = d6_fminf(d, temp[k]);
          temp[k] = t;
          score = t;
          if (WEIGHTED) score = (float)((double)t * fmax((double)weight_of(k), 1e-12));
        }
        const bool up = score > best;
        bk = up ? k : bk;
        best = up ? score : best;
      }
    }
    const float wmax = d6_wave_max(best);
    const unsigned long long tie = __ballot(best == wmax);
    const int wl = __builtin_ctzll(tie);
    const int wk = d6_readlane_i(bk, wl);
    Slot *sl = slots[r & 1];
    if (lane == 0) { sl[wave].val = wmax; sl[wave].idx = wk; }
    __syncthreads();
    const int src = lane < nwaves ? lane : 0;
    const float v2 = lane < nwaves ? sl[src].val : -__builtin_inff();
    const int i2 = sl[src].idx;
    const float bmax = d6_wave_max(v2);
    const unsigned long long tie2 = __ballot(v2 == bmax);
    old = d6_readlane_i(i2, __builtin_ctzll(tie2));
    if (h == 0) idxs[r] = old + vw.idx_add;
  }
}

// core/pcdet/ops/pointnet2/pointnet2_batch/src/cuda_utils.h:10-14 (same double formula, same libm)
int opt_n_threads_log2(int work_size) {
  int pow_2 = (int)(log((double)work_size) / log(2.0));
  if (pow_2 > 10) pow_2 = 10;
  if (pow_2 < 0) pow_2 = 0;
  return pow_2;
}

// experiments build: DET6D_FPS_NO_FASTW=1 keeps the exact-double S-FPS launch only (A/B)
static bool no_fastw() {
  static const bool off = det6d_env_int("DET6D_FPS_NO_FASTW", 0) != 0;
  return off;
}

template <bool W>
int launch_fps(int b, int n, int m, const float *xyz, const float *weights, float *temp, int *idx,
               const FpsView &vw, hipStream_t stream) {
  if (b < 0 || n <= 0 || m < 0) return DET6D_EINVAL;
  if (b == 0 || m == 0) return DET6D_OK;   // nothing to do (empty tensors may carry null pointers)
  if (!xyz || !idx || (W && !weights) || (!temp && !vw.init_temp)) return DET6D_EINVAL;
  const int log2s = opt_n_threads_log2(n);
  const int S = 1 << log2s;
  const int threads = S < 64 ? 64 : S;
  const int ppt = (n + S - 1) / S;
  dim3 grid(b), block(threads);
  // fat-thread kernels: N = SLOTS << LOG2T exactly, T <= S
  // S-FPS with a free workspace (det6d_fps_fused: the min-distances start at 1e10 implicitly, `temp` is scratch): the fp32
  // scoring launch, then the exact-double launch for the scenes it handed over (flag word = first word of the scene's scratch)
  int *flags = (W && vw.init_temp && temp && !no_fastw()) ? reinterpret_cast<int *>(temp) : nullptr;
#define FPS_FAT(LT, SL)                                                                        \
  do {                                                                                         \
    int lp = 0;                                                                                \
    while ((1 << lp) < ppt) ++lp;                                                              \
    if (flags) {                                                                               \
      /* the opt-in for the dynamic LDS belongs to the instantiation that is launched; a failed first launch must not be  */ \
      /* followed by the guarded one (it would read flag words nobody wrote)                                              */ \
      static const unsigned hogw = det6d_sampler_lds_hog(fps_fat_kernel<LT, SL, W, W>, 1024);  \
      hipLaunchKernelGGL((fps_fat_kernel<LT, SL, W, W>), grid, dim3(1 << LT), hogw, stream, n, m, log2s, \
                         lp, xyz, weights, temp, idx, vw, flags);                              \
      const int rc_first = det6d_check_launch("det6d_fps (fp32 scoring)");                     \
      if (rc_first != DET6D_OK) return rc_first;                                               \
      hipLaunchKernelGGL((fps_mem_kernel<W>), grid, dim3(S < 64 ? 64 : S), 0, stream, n, m, log2s, xyz, weights, temp, idx, vw, 1); \
    } else {                                                                                   \
      static const unsigned hog = det6d_sampler_lds_hog(fps_fat_kernel<LT, SL, W>, 1024);      \
      hipLaunchKernelGGL((fps_fat_kernel<LT, SL, W>), grid, dim3(1 << LT), hog, stream, n, m, log2s, \
                         lp, xyz, weights, temp, idx, vw, nullptr);                            \
    }                                                                                          \
    return det6d_check_launch("det6d_fps");                                                    \
  } while (0)
  if (n == S * ppt && (ppt & (ppt - 1)) == 0) {
    // thread counts picked by measurement on MI355X (scripts/gpu_fps_time.py): the scan is bound
    // by VALU issue (7 instructions per point), the argmax/broadcast chain by latency
    if (n == 16384) FPS_FAT(9, 32);
    if (n == 8192) FPS_FAT(9, 16);
    if (n == 4096) FPS_FAT(9, 8);
    if (n == 2048) FPS_FAT(7, 16);
    if (n == 1024) FPS_FAT(6, 16);
    if (n == 512) FPS_FAT(6, 8);
    if (n == 256) FPS_FAT(6, 4);
    if (n == 128) FPS_FAT(6, 2);
  }
#undef FPS_FAT
#define FPS_CASE(P)                                                                          \
  hipLaunchKernelGGL((fps_reg_kernel<P, W>), grid, block, 0, stream, n, m, log2s, xyz, weights, \
                     temp, idx, vw)
  if (ppt <= 1) FPS_CASE(1);
  else if (ppt <= 2) FPS_CASE(2);
  else if (ppt <= 4) FPS_CASE(4);
  else if (ppt <= 8) FPS_CASE(8);
  else if (ppt <= kMaxPPT) FPS_CASE(16);
  else {
    if (!temp) return DET6D_EINVAL;  // the memory-resident kernel needs its min-distance scratch
    hipLaunchKernelGGL((fps_mem_kernel<W>), grid, block, 0, stream, n, m, log2s, xyz, weights, temp,
                       idx, vw);
  }
#undef FPS_CASE
  return det6d_check_launch("det6d_fps");
}

FpsView dense_view(int n, int m) {
  FpsView vw;
  vw.xyz_bstride = (long long)n * 3;
  vw.w_bstride = n;
  vw.temp_bstride = n;
  vw.idx_bstride = m;
  vw.idx_add = 0;
  vw.init_temp = 0;
  vw.w_is_score = 0;
  vw.gamma = 1.0f;
  return vw;
}

}  // namespace

DET6D_API int det6d_fps(int b, int n, int m, const float *xyz, float *temp, int *idx,
                        det6d_stream_t stream) {
  return launch_fps<false>(b, n, m, xyz, nullptr, temp, idx, dense_view(n, m), (hipStream_t)stream);
}

DET6D_API int det6d_fps_weights(int b, int n, int m, const float *xyz, const float *weights,
                                float *temp, int *idx, det6d_stream_t stream) {
  return launch_fps<true>(b, n, m, xyz, weights, temp, idx, dense_view(n, m), (hipStream_t)stream);
}

// fps_cells.hip: exact spatially pruned D-FPS of 16384-point scenes (k-d regions + bounding-box skip test)
int det6d_fps_cells_launch(int b, int n, int m, int log2s, long long xyz_bstride, long long idx_bstride, int idx_add,
                           const float *xyz, int *perm, int *idx, hipStream_t stream);

#ifdef DET6D_EXPERIMENTS
// ---- profiling stand-ins (DET6D_FPS_STANDIN=1|2, never a result path; scripts/experiments/gpu_whatif.py): only in
// ---- libraries built with -DDET6D_EXPERIMENTS, never in the shipped one
// Same launch shape and duration as the SA1 sampler (one 512-thread workgroup per scene, ~1.35 us per round)
// but 1: holds 128 VGPRs and sleeps (register / occupancy footprint only), 2: keeps the vector ALU busy
// from a small register footprint.  Both write a strided index pattern so that the rest of the pass runs.
namespace {
template <int MODE>
__global__ __launch_bounds__(512) void fps_standin_kernel(int n, int m, int *idx, int idx_stride, int idx_add) {
  float r[MODE == 1 ? 120 : 8];
#pragma unroll
  for (int i = 0; i < (MODE == 1 ? 120 : 8); ++i) r[i] = threadIdx.x + i;
  for (int round = 0; round < m; ++round) {
    if (MODE == 1) {
      __builtin_amdgcn_s_sleep(40);   // 40 x 64 clocks ~ 1.1 us
    } else {
#pragma unroll
      for (int u = 0; u < 26; ++u)    // ~208 dependent-free VALU ops, like the real scan
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
    }
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < (MODE == 1 ? 120 : 8); ++i) { asm volatile("" : "+v"(r[i])); s += r[i]; }
  int *out = idx + (size_t)blockIdx.x * idx_stride;
  for (int j = threadIdx.x; j < m; j += blockDim.x) out[j] = (int)(((long long)j * n) / m) + idx_add + (s == -1.f);
}
}  // namespace
#endif

// fps_coop.hip: register-resident D-FPS of 32768 / 65536-point scenes by cooperating workgroups
bool det6d_fps_coop_handles(int n);
long long det6d_fps_coop_workspace_bytes(int b, int n);
int det6d_fps_cells_w_launch(int b, int n, int m, int log2s, long long xyz_bstride, long long idx_bstride, int idx_add,
                             const float *xyz, int *perm, int *idx, const float *weights, long long w_bstride, float gamma,
                             int w_is_score, hipStream_t stream);      // fps_cells.hip
bool det6d_fps_coop_fits_device(int n);      // fps_coop.hip: the current device holds one cooperative launch (>= 8 x parts CUs)
int det6d_fps_coop_launch(int b, int n, int m, int log2s, long long xyz_bstride, long long idx_bstride, int idx_add,
                          const float *xyz, void *workspace, int *idx, hipStream_t stream);

int det6d_fps_coop_status(int b, int n, const void *workspace, hipStream_t stream);
long long det6d_fps_coop_status_offset(int b, int n);

DET6D_API int det6d_fps_fused_status(int b, int n, const float *temp, long long temp_bytes, det6d_stream_t stream) {
  if (!temp || !det6d_fps_coop_handles(n)) return DET6D_OK;     // only the cooperative sampler can fail after its launch
  const char *ws = reinterpret_cast<const char *>(((uintptr_t)temp + 255) & ~(uintptr_t)255);
  const long long need = det6d_fps_coop_workspace_bytes(b, n);
  if (need <= 0 || temp_bytes - (ws - reinterpret_cast<const char *>(temp)) < need) return DET6D_OK;
  return det6d_fps_coop_status(b, n, ws, (hipStream_t)stream);
}

DET6D_API long long det6d_fps_fused_status_offset(int b, int n, const float *temp, long long temp_bytes) {
  if (!temp || !det6d_fps_coop_handles(n)) return -1;
  const char *ws = reinterpret_cast<const char *>(((uintptr_t)temp + 255) & ~(uintptr_t)255);
  const long long need = det6d_fps_coop_workspace_bytes(b, n);
  if (need <= 0 || temp_bytes - (ws - reinterpret_cast<const char *>(temp)) < need) return -1;
  return (ws - reinterpret_cast<const char *>(temp)) + det6d_fps_coop_status_offset(b, n);
}

DET6D_API long long det6d_fps_fused_workspace_bytes(int b, int n) {
  if (b <= 0 || n <= 0) return 0;
  long long bytes = (long long)b * n * 4;
  const long long coop = det6d_fps_coop_workspace_bytes(b, n);
  if (coop > 0 && coop + 256 > bytes) bytes = coop + 256;
  return bytes;
}

DET6D_API int det6d_fps_fused(int b, int n_total, int lo, int hi, int m, const float *xyz,
                              const float *scores, float gamma, float *temp, long long temp_bytes, int *idx,
                              int idx_stride, int idx_offset, int idx_bias, det6d_stream_t stream) {
  if (n_total <= 0 || lo < 0 || hi > n_total || hi <= lo || idx_stride < idx_offset + m) return DET6D_EINVAL;
  const int n = hi - lo;
  if (temp && temp_bytes < (long long)b * n * 4) return DET6D_EINVAL;
  FpsView vw;
  vw.xyz_bstride = (long long)n_total * 3;
  vw.w_bstride = n_total;
  vw.temp_bstride = n;
  vw.idx_bstride = idx_stride;
  vw.idx_add = lo + idx_bias;
  vw.init_temp = 1;
  vw.w_is_score = 1;
  vw.gamma = gamma;
  const float *x = xyz ? xyz + (size_t)lo * 3 : nullptr;
  int *out = idx ? idx + idx_offset : nullptr;
  // (round 6) S-FPS of 16384- / 4096-point clouds in the multi-pick form (fps_seq.hip: fps_seq_w_kernel on the k-d regions of
  // fps_cells.hip), behind it the guarded exact-double launch for scenes that hold a weight below 1e-12; DET6D_FPS_SEQW=0
  // (knobs / experiments build): the one-pick fat-thread kernel of rounds 2-5
  static const int seqw = det6d_env_int("DET6D_FPS_SEQW", 1);
  if (scores && seqw && temp && x && out && b > 0 && m > 0 && m <= n && (n == 16384 || n == 4096) && !no_fastw()) {
    const int log2s = opt_n_threads_log2(n);
    const int rc = det6d_fps_cells_w_launch(b, n, m, log2s, vw.xyz_bstride, vw.idx_bstride, lo + idx_bias, x, reinterpret_cast<int *>(temp),
                                            out, scores + lo, vw.w_bstride, gamma, 1, (hipStream_t)stream);
    if (rc != DET6D_OK) return rc;
    const int S = 1 << log2s;
    hipLaunchKernelGGL((fps_mem_kernel<true>), dim3(b), dim3(S < 64 ? 64 : S), 0, (hipStream_t)stream, n, m, log2s, x, scores + lo, temp, out,
                       vw, 1);
    return det6d_check_launch("det6d_fps (score-weighted, guarded exact launch)");
  }
  if (scores) return launch_fps<true>(b, n, m, x, scores + lo, temp, out, vw, (hipStream_t)stream);
#ifdef DET6D_EXPERIMENTS
  static const int standin = det6d_env_int("DET6D_FPS_STANDIN", 0);
  if (standin && n == 16384 && out) {
    if (standin == 1) hipLaunchKernelGGL(fps_standin_kernel<1>, dim3(b), dim3(512), 0, (hipStream_t)stream, n, m, out, idx_stride, lo + idx_bias);
    else hipLaunchKernelGGL(fps_standin_kernel<2>, dim3(b), dim3(512), 0, (hipStream_t)stream, n, m, out, idx_stride, lo + idx_bias);
    return det6d_check_launch("det6d_fps_fused(stand-in)");
  }
#endif
  // 32768 / 65536 points: the scene is held in registers by 2 / 4 cooperating workgroups (fps_coop.hip) when the caller
  // supplied the workspace det6d_fps_fused_workspace_bytes asks for; otherwise the memory-resident kernel below, 100x
  // slower, same picks
  if (temp && x && out && b > 0 && m > 0 && det6d_fps_coop_handles(n) && det6d_fps_coop_fits_device(n)) {
    char *ws = reinterpret_cast<char *>(((uintptr_t)temp + 255) & ~(uintptr_t)255);
    const long long avail = temp_bytes - (ws - reinterpret_cast<char *>(temp));
    const long long need = det6d_fps_coop_workspace_bytes(b, n);
    if (need > 0 && avail >= need)
      return det6d_fps_coop_launch(b, n, m, opt_n_threads_log2(n), vw.xyz_bstride, vw.idx_bstride, lo + idx_bias, x, ws, out, (hipStream_t)stream);
  }
  // 16384 / 4096 points: the multi-pick sampler of fps_seq.hip on the k-d regions of fps_cells.hip (16 / 4 points per lane in
  // registers, one bounding box per wave; the permutation lives in `temp`, which is free because the min-distances start at
  // 1e10 implicitly): same picks bit for bit, 0.51 vs 1.35 us per pick of the plain fat-thread kernel at 16384 points
  if (temp && x && out && b > 0 && m > 0 && (n == 16384 || n == 4096))
    return det6d_fps_cells_launch(b, n, m, opt_n_threads_log2(n), vw.xyz_bstride, vw.idx_bstride, lo + idx_bias, x,
                                  reinterpret_cast<int *>(temp), out, (hipStream_t)stream);
  return launch_fps<false>(b, n, m, x, nullptr, temp, out, vw, (hipStream_t)stream);
}
