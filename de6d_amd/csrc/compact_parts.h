// compact_parts.h — the rule that cuts a centre's rows into power-of-two parts (compact.hip), shared with the ball-query
// kernel that counts the parts of its own 256 centres while it still holds their hit counts (ball_query_grid.hip).
#pragma once

constexpr int kCompactClasses = 6;

// classes (bit c <-> s = 32 >> c) the rows of a centre with `cnt` hits are placed in.
//   split = 0: one part, the next power of two >= max(cnt, smin);
//   split = g > 0: up to g hits the same single part; beyond, ceil(cnt / g) * g rows, cut along their binary digits
//              into parts of descending size (20 rows = 16 + 4: slots 0..15 form a class-16 group, slots 16..19 a
//              class-4 group); the pooled value of the centre is the maximum over its parts, combined by an integer
//              atomic max on the (non-negative, post-ReLU) outputs (bit 29 of crow_c marks such rows; the pooled
//              buffer is zeroed first).  smin = 1, g = 4: singles and pairs are rows of their own, no atomics for them.
__device__ __forceinline__ int d6_compact_parts_of(int cnt, int ns, int smin, int split_tol, int *rows_out) {
  const int k = cnt < 1 ? 1 : (cnt > ns ? ns : cnt);
  const int split = split_tol & 0xff, tol = split_tol >> 8;   // tol t > 0: one power-of-two part when it wastes <= 1/t of its rows
  int p2 = smin;
  while (p2 < k) p2 <<= 1;
  int rows;
  if (split > 0 && k > split && !(tol > 0 && (p2 - k) * tol <= p2)) {
    rows = (k + split - 1) / split * split;
  } else {
    rows = p2;
  }
  *rows_out = rows;
  int mask = 0;
#pragma unroll
  for (int c = 0; c < kCompactClasses; ++c)
    if (rows & (32 >> c)) mask |= 1 << c;
  return mask;
}

// what the list builder needs from whoever counts: per block of 256 centres, parts per class [0..5] and information rows [6]
// -> table[block * 7 + c], table = hdr + 16 (compact.hip: compact_place_kernel reads it)
struct CompactCountArgs {
  int ns, smin, split;
  int *table;                // nullptr: no counting
};

// host side (compact.hip): validates one group's (ns, smin, split) and clamps split to ns; DET6D_OK or DET6D_EINVAL
int det6d_compact_check_group(int ns, int smin, int *split);
// split with the experiments build's DET6D_COMPACT_TOL folded in (bits 8..): what the kernels' parts_of() takes
int det6d_compact_split_tol(int split);
