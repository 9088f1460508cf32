// common.h — shared host/device helpers for libdet6d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/det6d_math.h"
#include "../../include/det6d_ops.h"

#define DET6D_API extern "C" __attribute__((visibility("default")))

// Record the failing HIP call for det6d_last_error(); never exit() (the reference does:
// sampling_gpu.cu:261-265).
void det6d_set_error(const char *what, hipError_t err);

static inline int det6d_check_launch(const char *what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    det6d_set_error(what, err);
    return DET6D_ELAUNCH;
  }
  return DET6D_OK;
}

static inline int det6d_divup(int a, int b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set it once per (call site, device) — a
// process-wide "done" flag would leave a second device of the process at the default limit.  The race of two host threads on
// the remembered device only repeats an idempotent call.
#define DET6D_MAX_DYNAMIC_LDS(kernel, bytes)                                                                       \
  do {                                                                                                              \
    static int d6_attr_dev_ = -1;                                                                                   \
    int d6_dev_ = 0;                                                                                                \
    if (hipGetDevice(&d6_dev_) == hipSuccess && d6_attr_dev_ != d6_dev_) {                                          \
      hipFuncSetAttribute((const void *)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));        \
      d6_attr_dev_ = d6_dev_;                                                                                       \
    }                                                                                                               \
  } while (0)

// Environment switches.  Two kinds:
//  * det6d_switch_*: read by the SHIPPED library: DET6D_FPS_COOP_FAST (fps_coop.hip) and nothing else (DET6D_DENSE_ROWS is
//    read by the Python side).
//  * det6d_env_*: alternative kernel routes (all result-preserving: DET6D_LINEAR_NO_FAST, DET6D_CHAIN_LDS,
//    DET6D_CHAIN_NO_WIDE, DET6D_ROWS_RB, DET6D_GROUP_STREAM, DET6D_GROUP_PRE), tile sweeps, timing hooks and stand-ins used by
//    scripts/experiments and by the route tests.  Compiled to their default unless the library is built with
//    -DDET6D_EXPERIMENTS (python -m de6d_amd._build --experiments): the shipped library ignores those variables.
static inline int det6d_switch_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v ? atoi(v) : dflt;
}
static inline bool det6d_switch_set(const char *name) { return getenv(name) != nullptr; }
// -DDET6D_KNOBS (python -m de6d_amd._build --knobs -> libdet6d_hip_knobs.so, loaded with DET6D_KNOBS_LIB=1): the SHIPPED kernels
// with the det6d_env_* route / tile switches live and nothing else — no timers, statistics or debug hooks.  Round 6: an A/B
// inside the experiments library said +10 % for two settings that are worth +0.5 % in the shipped one (its instrumented
// one-pass group kernel is the slow side of that comparison); A/B runs of routes belong in this flavour.
#if defined(DET6D_EXPERIMENTS) || defined(DET6D_KNOBS)
static inline int det6d_env_int(const char *name, int dflt) { return det6d_switch_int(name, dflt); }
static inline bool det6d_env_set(const char *name) { return det6d_switch_set(name); }
#else
static inline int det6d_env_int(const char *, int dflt) { return dflt; }
static inline bool det6d_env_set(const char *) { return false; }
#endif
#ifdef DET6D_EXPERIMENTS
#define D6_DBG_IS(v) (dbg == (v))
// DET6D_DBG_POISON_LDS=<pattern>: fill the LDS of every CU before a sampler kernel (fps_seq.hip; tests only)
void det6d_dbg_poison_lds_hook(hipStream_t stream);
// DET6D_GEMM_PRIO=1 (experiments build): every wave of the GEMM family raises its issue priority (s_setprio 3) — does the matrix
// stream lose issue slots to the vector instructions of the kernels beside it?  (scripts/r05/gpu_t34.sh)
#define D6_GEMM_PRIO_DECL __device__ int d6_gemm_prio_flag = 0;
#define D6_GEMM_PRIO_APPLY() do { if (d6_gemm_prio_flag) __builtin_amdgcn_s_setprio(3); } while (0)
#define D6_GEMM_PRIO_HOST()                                                                                  \
  do {                                                                                                       \
    static bool d6_p_ = false;                                                                               \
    if (!d6_p_) {                                                                                            \
      d6_p_ = true;                                                                                          \
      const int d6_v_ = det6d_env_int("DET6D_GEMM_PRIO", 0);                                                 \
      if (d6_v_) (void)hipMemcpyToSymbol(HIP_SYMBOL(d6_gemm_prio_flag), &d6_v_, sizeof(int));                \
    }                                                                                                        \
  } while (0)
#else
#define D6_GEMM_PRIO_DECL
#define D6_GEMM_PRIO_APPLY() do {} while (0)
#define D6_GEMM_PRIO_HOST() do {} while (0)
#define D6_DBG_IS(v) false
#endif

// squared distance with the contraction order documented in oracle/det6d_oracle.c
__device__ __forceinline__ float d6_sqdist(float dx, float dy, float dz) {
  return D6_FMA(dz, dz, D6_FMA(dx, dx, dy * dy));
}

// ---- wave64 cross-lane helpers (DPP, no LDS) ---------------------------------------------------
// One v_max_f32_dpp per reduction step.  (The builtin form, update_dpp + fmaxf, expands to mov +
// mov_dpp + two canonicalising max per step.)  Inline asm gets no hazard padding from the compiler: a
// DPP read of a VGPR written by the previous VALU op needs 2 wait states, hence the s_nop 1.
// Inputs must not be NaN.

// max over the 64 lanes of a wave; uniform result (read from lane 63)
__device__ __forceinline__ float d6_wave_max(float v) {
  float t;
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      : "=&v"(t)
      : "v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63));
}
__device__ __forceinline__ float d6_wave_min(float v) { return -d6_wave_max(-v); }

// max over lanes 0..15 (one DPP row); uniform result (read from lane 0)
__device__ __forceinline__ float d6_row_max16(float v) {
  float t;
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      : "=&v"(t)
      : "v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 0));
}

// v_min_f32 == fminf without the canonicalising v_max the builtin adds for possible signalling NaNs
__device__ __forceinline__ float d6_vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// one-instruction max / ReLU (the C forms cost a canonicalising v_max or a compare + select first); no NaN inputs.
// On gfx950 every VALU instruction beside fp32 MFMAs is matrix time lost, hence these.
__device__ __forceinline__ float d6_vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float d6_relu(float v) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
  return r;
}

// The samplers are latency chains of a few hundred dependent vector-ALU instructions per round on ONE workgroup
// per scene, co-resident with GEMM waves whose fp32 MFMAs occupy the same vector ALU for 64 cycles each: at the
// default priority every sampler instruction queues behind an MFMA.  Raised wave priority lets the chain's
// instructions issue first (the GEMM waves lose a few issue slots per round, the sampler leaves the CU sooner).
__device__ __forceinline__ void d6_sampler_priority() {
#ifndef DET6D_NO_SAMPLER_PRIO
  __builtin_amdgcn_s_setprio(3);
#endif
}

// Dynamic LDS a sampler launch asks for ON TOP of its static use so that no workgroup that needs more than a few
// KB of LDS (every GEMM-family kernel) becomes co-resident with it: beside fp32-MFMA waves the sampler's dependent
// vector-ALU chain runs 5-6x slower (each instruction queues behind a 64-cycle MFMA; scripts/gpu_fps_interf.py:
// 3.4 -> 6.5 ms in-kernel for the SA1 sampler).  Measured in the two-stage pipeline the co-resident GEMM waves are
// still a net gain (9640 scenes/s without the reservation, 9235 with 8 KB left to others, 9670 with 40 KB), so the
// reservation is OFF by default; DET6D_FPS_LDS_HOG=<KB left to other workgroups> turns it on.
template <typename KernelT>
static inline unsigned det6d_sampler_lds_hog(KernelT kernel, unsigned static_bytes) {
  static const int keep_kb = det6d_env_int("DET6D_FPS_LDS_HOG", 0);   // LDS left to others (KB); 0 = no reservation
  if (keep_kb <= 0) return 0u;
  const unsigned total = 160u * 1024u, keep = (unsigned)keep_kb * 1024u;
  if (static_bytes + keep >= total) return 0u;
  const unsigned dyn = total - keep - static_bytes;
  hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  return dyn;
}

__device__ __forceinline__ float d6_readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ int d6_readlane_i(int v, int lane) {
  return __builtin_amdgcn_readlane(v, lane);
}
