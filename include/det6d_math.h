/*
 * det6d_math.h — deterministic fp32 math shared by the HIP kernels and the CPU oracle.
 *
 * Why this exists: the reference calls libm (`cos/sin/atan2` in
 * core/pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:51-102, `torch.sigmoid/exp` in
 * core/pcdet/utils/box_coder_utils.py:622-680).  Device libm (ocml) and glibc differ in the
 * last bits, which would make "bit-exact NMS keep masks" a matter of luck.  Every
 * transcendental on the hot path therefore goes through the routines below, which use only
 * IEEE-754 correctly rounded primitives (+ - * / fma, int<->float casts, bit moves) in a
 * fixed order, so gcc (x86-64, -ffp-contract=off -mfma) and hipcc (gfx950, -ffp-contract=off,
 * correctly rounded divide) produce identical bits.
 *
 * Accuracy (checked in tests/test_math.py against numpy float64): expf <= 2 ulp,
 * sinf/cosf <= 2 ulp for |x| <= 1e3 rad, atan2f <= 3 ulp, logf <= 2 ulp.
 * Polynomials are the classic Cephes single-precision minimax sets.
 */
#ifndef DET6D_MATH_H
#define DET6D_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define D6_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#include <string.h>
#define D6_HD static inline
#endif

/* single rounding fused multiply-add on both sides (v_fma_f32 / vfmadd) */
#define D6_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

D6_HD float d6_bits2f(uint32_t u) {
#if defined(__HIPCC__)
  return __builtin_bit_cast(float, u);
#else
  float f; memcpy(&f, &u, 4); return f;
#endif
}
D6_HD uint32_t d6_f2bits(float f) {
#if defined(__HIPCC__)
  return __builtin_bit_cast(uint32_t, f);
#else
  uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}

D6_HD float d6_fabsf(float x) { return d6_bits2f(d6_f2bits(x) & 0x7fffffffu); }

/* fminf/fmaxf with the CUDA/C99 rule "if one operand is NaN return the other". */
#if defined(__HIP_DEVICE_COMPILE__)
/* v_min_f32 / v_max_f32 implement exactly this rule */
D6_HD float d6_fminf(float a, float b) { return __builtin_fminf(a, b); }
D6_HD float d6_fmaxf(float a, float b) { return __builtin_fmaxf(a, b); }
#else
D6_HD float d6_fminf(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
D6_HD float d6_fmaxf(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
#endif

/* round-to-nearest-even to an integer valued float, |x| < 2^22 */
D6_HD float d6_rintf_small(float x) {
  const float magic = 12582912.0f; /* 1.5 * 2^23 */
  float t = x + magic;
  return t - magic;
}

/* e^x */
D6_HD float d6_expf(float x) {
  if (x != x) return x;
  if (x > 88.72283905206835f) return d6_bits2f(0x7f800000u);
  if (x < -103.97f) return 0.0f;
  float n = d6_rintf_small(x * 1.44269504088896341f);
  float r = D6_FMA(n, -0.693359375f, x);
  r = D6_FMA(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = D6_FMA(p, r, 1.3981999507e-3f);
  p = D6_FMA(p, r, 8.3334519073e-3f);
  p = D6_FMA(p, r, 4.1665795894e-2f);
  p = D6_FMA(p, r, 1.6666665459e-1f);
  p = D6_FMA(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  float y = D6_FMA(p, r2, r) + 1.0f;
  /* scale by 2^n in two steps so that subnormal results and n = 128 stay exact */
  int ni = (int)n;
  int n1 = ni / 2, n2 = ni - n1;
  float s1 = d6_bits2f((uint32_t)(n1 + 127) << 23);
  float s2 = d6_bits2f((uint32_t)(n2 + 127) << 23);
  return (y * s1) * s2;
}

/* natural log, x > 0 finite normal or subnormal; returns -inf for 0, NaN for x < 0 */
D6_HD float d6_logf(float x) {
  if (x != x) return x;
  if (x < 0.0f) return d6_bits2f(0x7fc00000u);
  if (x == 0.0f) return d6_bits2f(0xff800000u);
  uint32_t u = d6_f2bits(x);
  if (u == 0x7f800000u) return x;
  int e = 0;
  if (u < 0x00800000u) { x = x * 8388608.0f; u = d6_f2bits(x); e = -23; }
  e += (int)(u >> 23) - 126;
  float m = d6_bits2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
  if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float p = 7.0376836292e-2f;
  p = D6_FMA(p, m, -1.1514610310e-1f);
  p = D6_FMA(p, m, 1.1676998740e-1f);
  p = D6_FMA(p, m, -1.2420140846e-1f);
  p = D6_FMA(p, m, 1.4249322787e-1f);
  p = D6_FMA(p, m, -1.6668057665e-1f);
  p = D6_FMA(p, m, 2.0000714765e-1f);
  p = D6_FMA(p, m, -2.4999993993e-1f);
  p = D6_FMA(p, m, 3.3333331174e-1f);
  float fe = (float)e;
  float y = (m * z) * p;
  y = D6_FMA(fe, -2.12194440e-4f, y);
  y = D6_FMA(z, -0.5f, y);
  float r = m + y;
  r = D6_FMA(fe, 0.693359375f, r);
  return r;
}

/* 1 / (1 + e^-x)   (torch.sigmoid restated; pointnet2_modules.py:419, box_coder_utils.py:628) */
D6_HD float d6_sigmoidf(float x) { return 1.0f / (1.0f + d6_expf(-x)); }

/* sigmoid(x) ** gamma as used for the s-fps weights (pointnet2_modules.py:415-424).
 * gamma == 1 is the identity (torch.pow special-cases it the same way). */
D6_HD float d6_sigmoid_powf(float x, float gamma) {
  float s = d6_sigmoidf(x);
  if (gamma == 1.0f) return s;
  if (s == 0.0f) return gamma > 0.0f ? 0.0f : (gamma == 0.0f ? 1.0f : d6_bits2f(0x7f800000u));
  return d6_expf(gamma * d6_logf(s));
}

/* sin and cos of x (radians). Octant reduction with a 3-term Cody-Waite split of pi/4. */
D6_HD void d6_sincosf(float x, float *s_out, float *c_out) {
  if (x != x || d6_fabsf(x) > 8388608.0f) {
    /* NaN/inf -> NaN; huge finite arguments are outside the supported range (yaw is O(2*pi)) */
    float bad = (x != x || d6_f2bits(d6_fabsf(x)) == 0x7f800000u) ? d6_bits2f(0x7fc00000u) : 0.0f;
    *s_out = bad; *c_out = (bad != bad) ? bad : 1.0f;
    return;
  }
  float ax = d6_fabsf(x);
  int j = (int)(ax * 1.27323954473516f); /* 4/pi */
  j = (j + 1) & ~1;                      /* map to even octant count */
  float y = (float)j;
  float r = D6_FMA(y, -0.78515625f, ax);
  r = D6_FMA(y, -2.4187564849853515625e-4f, r);
  r = D6_FMA(y, -3.77489497744594108e-8f, r);
  float z = r * r;
  float ps = -1.9515295891e-4f;
  ps = D6_FMA(ps, z, 8.3321608736e-3f);
  ps = D6_FMA(ps, z, -1.6666654611e-1f);
  float sr = D6_FMA(ps * z, r, r);
  float pc = 2.443315711809948e-5f;
  pc = D6_FMA(pc, z, -1.388731625493765e-3f);
  pc = D6_FMA(pc, z, 4.166664568298827e-2f);
  float cr = D6_FMA(pc * z, z, D6_FMA(z, -0.5f, 1.0f));
  int q = (j >> 1) & 3;
  float sv, cv;
  if (q == 0) { sv = sr; cv = cr; }
  else if (q == 1) { sv = cr; cv = -sr; }
  else if (q == 2) { sv = -sr; cv = -cr; }
  else { sv = -cr; cv = sr; }
  if (x < 0.0f) sv = -sv;
  *s_out = sv; *c_out = cv;
}
D6_HD float d6_sinf(float x) { float s, c; d6_sincosf(x, &s, &c); return s; }
D6_HD float d6_cosf(float x) { float s, c; d6_sincosf(x, &s, &c); return c; }

/* atan of a non-negative finite or infinite argument */
D6_HD float d6_atanf_pos(float x) {
  float y0;
  if (x > 2.414213562373095f) { y0 = 1.5707963267948966f; x = -1.0f / x; }
  else if (x > 0.4142135623730950f) { y0 = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
  else { y0 = 0.0f; }
  float z = x * x;
  float p = 8.05374449538e-2f;
  p = D6_FMA(p, z, -1.38776856032e-1f);
  p = D6_FMA(p, z, 1.99777106478e-1f);
  p = D6_FMA(p, z, -3.33329491539e-1f);
  float y = D6_FMA(p * z, x, x);
  return y0 + y;
}

/* atan2(y, x), C99 quadrant conventions for zeros; NaN in -> NaN out */
D6_HD float d6_atan2f(float y, float x) {
  if (x != x || y != y) return d6_bits2f(0x7fc00000u);
  const float PI = 3.14159265358979323846f;
  const float PIO2 = 1.5707963267948966f;
  int xneg = (d6_f2bits(x) >> 31) != 0;
  int yneg = (d6_f2bits(y) >> 31) != 0;
  float ax = d6_fabsf(x), ay = d6_fabsf(y);
  float r;
  if (ay == 0.0f) {
    r = xneg ? PI : 0.0f;
  } else if (ax == 0.0f) {
    r = PIO2;
  } else {
    uint32_t inf = 0x7f800000u;
    if (d6_f2bits(ax) == inf && d6_f2bits(ay) == inf) r = xneg ? 2.356194490192345f : 0.7853981633974483f;
    else if (d6_f2bits(ax) == inf) r = xneg ? PI : 0.0f;
    else if (d6_f2bits(ay) == inf) r = PIO2;
    else {
      float a = d6_atanf_pos(ay / ax);
      r = xneg ? (PI - a) : a;
    }
  }
  return yneg ? -r : r;
}

#endif /* DET6D_MATH_H */
