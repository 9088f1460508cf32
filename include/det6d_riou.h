/* det6d_riou.h — the rotated-box intersection of the KITTI evaluator, shared (like det6d_geom.h) by the
 * HIP kernels and the CPU oracle so both follow the same float32 operation sequence.
 *
 * Algorithm and operation order follow the reference's numba.cuda device functions
 * (core/pcdet/datasets/kitti/kitti_object_eval_python/rotate_iou.py):
 *   rbbox_to_corners :207-231, point_in_quadrilateral :163-181, line_segment_intersection :73-116,
 *   quadrilateral_intersection :184-204, sort_vertex_in_convex_polygon :32-70, area / trangle_area :16-29,
 *   inter :234-247, devRotateIoUEval :250-262
 * with numba's typing: float32 geometry, float64 area accumulation and final ratio.  No fma contraction
 * (the reference's NVVM build may contract; parity with it is therefore tolerance-level, 1e-5).
 * cos / sin come from det6d_math.h.
 */
#ifndef DET6D_RIOU_H_
#define DET6D_RIOU_H_

#include "det6d_math.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define d6_sqrtf_rn(x) __fsqrt_rn(x)
#else
#include <math.h>
#define d6_sqrtf_rn(x) sqrtf(x)
#endif

/* box: [cx, cy, x extent, y extent, angle (clockwise positive)] -> 4 corners (x0,y0,..,x3,y3) */
D6_HD void d6_riou_corners(const float *box, float *c) {
  float sn, cs;
  d6_sincosf(box[4], &sn, &cs);
  const float hx = box[2] / 2.f, hy = box[3] / 2.f;
  const float lx[4] = {-hx, -hx, hx, hx}, ly[4] = {-hy, hy, hy, -hy};
  for (int i = 0; i < 4; ++i) {
    c[2 * i] = cs * lx[i] + sn * ly[i] + box[0];
    c[2 * i + 1] = -sn * lx[i] + cs * ly[i] + box[1];
  }
}

D6_HD int d6_riou_inside(float px, float py, const float *q) {
  const float ab0 = q[2] - q[0], ab1 = q[3] - q[1];
  const float ad0 = q[6] - q[0], ad1 = q[7] - q[1];
  const float ap0 = px - q[0], ap1 = py - q[1];
  const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
  const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
  return abab >= abap && abap >= 0.f && adad >= adap && adap >= 0.f;
}

/* edge i of p1 against edge j of p2 */
D6_HD int d6_riou_cross(const float *p1, const float *p2, int i, int j, float *out) {
  const float a0 = p1[2 * i], a1 = p1[2 * i + 1];
  const float b0 = p1[2 * ((i + 1) & 3)], b1 = p1[2 * ((i + 1) & 3) + 1];
  const float c0 = p2[2 * j], c1 = p2[2 * j + 1];
  const float d0 = p2[2 * ((j + 1) & 3)], d1 = p2[2 * ((j + 1) & 3) + 1];
  const float ba0 = b0 - a0, ba1 = b1 - a1, da0 = d0 - a0, ca0 = c0 - a0, da1 = d1 - a1, ca1 = c1 - a1;
  const int acd = da1 * ca0 > ca1 * da0;
  const int bcd = (d1 - b1) * (c0 - b0) > (c1 - b1) * (d0 - b0);
  if (acd == bcd) return 0;
  const int abc = ca1 * ba0 > ba1 * ca0;
  const int abd = da1 * ba0 > ba1 * da0;
  if (abc == abd) return 0;
  const float dc0 = d0 - c0, dc1 = d1 - c1;
  const float abba = a0 * b1 - b0 * a1, cddc = c0 * d1 - d0 * c1;
  const float dh = ba1 * dc0 - ba0 * dc1;
  out[0] = (abba * dc0 - ba0 * cddc) / dh;
  out[1] = (abba * dc1 - ba1 * cddc) / dh;
  return 1;
}

/* area of the intersection of two rotated rectangles (float64 accumulation of float32 triangle areas) */
D6_HD double d6_riou_inter(const float *box1, const float *box2) {
  float c1[8], c2[8], pts[48], key[24];  /* <= 8 corner hits + 16 edge crossings */
  d6_riou_corners(box1, c1);
  d6_riou_corners(box2, c2);
  int n = 0;
  for (int i = 0; i < 4; ++i) {
    if (d6_riou_inside(c1[2 * i], c1[2 * i + 1], c2)) { pts[2 * n] = c1[2 * i]; pts[2 * n + 1] = c1[2 * i + 1]; ++n; }
    if (d6_riou_inside(c2[2 * i], c2[2 * i + 1], c1)) { pts[2 * n] = c2[2 * i]; pts[2 * n + 1] = c2[2 * i + 1]; ++n; }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float t[2];
      if (d6_riou_cross(c1, c2, i, j, t)) { pts[2 * n] = t[0]; pts[2 * n + 1] = t[1]; ++n; }
    }
  if (n == 0) return 0.0;
  /* order the vertices around their centroid by the pseudo-angle of the reference */
  float mx = 0.f, my = 0.f;
  for (int i = 0; i < n; ++i) { mx += pts[2 * i]; my += pts[2 * i + 1]; }
  mx = mx / (float)n; my = my / (float)n;
  for (int i = 0; i < n; ++i) {
    float vx = pts[2 * i] - mx, vy = pts[2 * i + 1] - my;
    const float d = d6_sqrtf_rn(vx * vx + vy * vy);
    vx = vx / d; vy = vy / d;
    if (vy < 0.f) vx = -2.f - vx;
    key[i] = vx;
  }
  for (int i = 1; i < n; ++i) {
    if (key[i - 1] > key[i]) {
      const float k = key[i], tx = pts[2 * i], ty = pts[2 * i + 1];
      int j = i;
      while (j > 0 && key[j - 1] > k) {
        key[j] = key[j - 1]; pts[2 * j] = pts[2 * j - 2]; pts[2 * j + 1] = pts[2 * j - 1];
        --j;
      }
      key[j] = k; pts[2 * j] = tx; pts[2 * j + 1] = ty;
    }
  }
  double area = 0.0;
  for (int i = 0; i < n - 2; ++i) {
    const float tri = (pts[0] - pts[2 * i + 4]) * (pts[2 * i + 3] - pts[2 * i + 5]) -
                      (pts[1] - pts[2 * i + 5]) * (pts[2 * i + 2] - pts[2 * i + 4]);
    const double half = (double)tri / 2.0;
    area += half < 0.0 ? -half : half;
  }
  return area;
}

/* criterion: -1 IoU, 0 inter / area(box1), 1 inter / area(box2), 2 intersection area; result as stored (float32) */
D6_HD float d6_riou_eval(const float *box1, const float *box2, int criterion) {
  const float a1 = box1[2] * box1[3], a2 = box2[2] * box2[3];
  const double in = d6_riou_inter(box1, box2);
  if (criterion == -1) return (float)(in / ((double)(a1 + a2) - in));
  if (criterion == 0) return (float)(in / (double)a1);
  if (criterion == 1) return (float)(in / (double)a2);
  return (float)in;
}

#endif /* DET6D_RIOU_H_ */
