// common.h — shared host/device helpers for libdet6d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// squared distance with the contraction order documented in oracle/det6d_oracle.c
__device__ __forceinline__ float d6_sqdist(float dx, float dy, float dz) {
  return D6_FMA(dz, dz, D6_FMA(dx, dx, dy * dy));
}

// ---- wave64 cross-lane helpers (DPP, no LDS) ------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float d6_dpp(float keep, float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, v),
                                         CTRL, ROW_MASK, 0xF, false));
}

// max over the 64 lanes of a wave; the result is uniform (returned from lane 63 via readlane).
// Inputs must not be NaN.
__device__ __forceinline__ float d6_wave_max(float v) {
  v = fmaxf(v, d6_dpp<0xB1>(v, v));        // quad_perm [1,0,3,2]
  v = fmaxf(v, d6_dpp<0x4E>(v, v));        // quad_perm [2,3,0,1]
  v = fmaxf(v, d6_dpp<0x141>(v, v));       // row_half_mirror
  v = fmaxf(v, d6_dpp<0x140>(v, v));       // row_mirror   -> every lane holds its row's max
  v = fmaxf(v, d6_dpp<0x142, 0xA>(v, v));  // row_bcast:15 -> rows 1,3
  v = fmaxf(v, d6_dpp<0x143, 0xC>(v, v));  // row_bcast:31 -> rows 2,3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float d6_readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ int d6_readlane_i(int v, int lane) {
  return __builtin_amdgcn_readlane(v, lane);
}
