/*
 * det6d_geom.h — rotated-rectangle (BEV) overlap / IoU arithmetic shared by the HIP kernels and
 * the CPU oracle.  One source so that both sides run the identical fp32 operation sequence
 * (compile with -ffp-contract=off); the sequence itself restates the reference's
 * core/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:59-229 (== iou3d_nms_kernel.cu:35-234 modulo nvcc's
 * FMA contraction) over plain float pairs instead of its Point class, and is pinned against a
 * build of that very file (oracle/_ref, tests/test_oracle_ref.py).
 */
#ifndef DET6D_GEOM_H
#define DET6D_GEOM_H
#include "det6d_math.h"

/* ------------------------------------------------------------------------------------------
 * Rotated BEV overlap. Follows core/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:59-229
 * (== iou3d_nms_kernel.cu:35-234 modulo nvcc's FMA contraction), written over plain float
 * pairs instead of the reference's Point class.
 * ---------------------------------------------------------------------------------------- */
#define IOU_EPS 1e-8f

D6_HD float d6g_fmin(float a, float b) { return a > b ? b : a; } /* iou3d_cpu.cpp:30-32 */
D6_HD float d6g_fmax(float a, float b) { return a > b ? a : b; } /* iou3d_cpu.cpp:34-36 */

/* cross(p1, p2, p0), iou3d_cpu.cpp:63-65 */
D6_HD float d6g_cross3(const float *p1, const float *p2, const float *p0) {
  return (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1]);
}

/* intersection(p1, p0, q1, q0, ans), iou3d_cpu.cpp:87-116 */
D6_HD int d6g_seg_intersection(const float *p1, const float *p0, const float *q1, const float *q0,
                            float *ans) {
  /* check_rect_cross(p0, p1, q0, q1), iou3d_cpu.cpp:67-73 */
  const int rect = d6g_fmin(p0[0], p1[0]) <= d6g_fmax(q0[0], q1[0]) &&
                   d6g_fmin(q0[0], q1[0]) <= d6g_fmax(p0[0], p1[0]) &&
                   d6g_fmin(p0[1], p1[1]) <= d6g_fmax(q0[1], q1[1]) &&
                   d6g_fmin(q0[1], q1[1]) <= d6g_fmax(p0[1], p1[1]);
  if (!rect) return 0;
  const float s1 = d6g_cross3(q0, p1, p0);
  const float s2 = d6g_cross3(p1, q1, p0);
  const float s3 = d6g_cross3(p0, q1, q0);
  const float s4 = d6g_cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  const float s5 = d6g_cross3(q1, p1, p0);
  if (d6_fabsf(s5 - s1) > IOU_EPS) {
    ans[0] = (s5 * q0[0] - s1 * q1[0]) / (s5 - s1);
    ans[1] = (s5 * q0[1] - s1 * q1[1]) / (s5 - s1);
  } else {
    const float a0 = p0[1] - p1[1], b0 = p1[0] - p0[0], c0 = p0[0] * p1[1] - p1[0] * p0[1];
    const float a1 = q0[1] - q1[1], b1 = q1[0] - q0[0], c1 = q0[0] * q1[1] - q1[0] * q0[1];
    const float D = a0 * b1 - a1 * b0;
    ans[0] = (b0 * c1 - b1 * c0) / D;
    ans[1] = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

/* check_in_box2d, iou3d_cpu.cpp:75-85; cos(-t) = cos t, sin(-t) = -sin t exactly */
D6_HD int d6g_point_in_box2d(const float *box, float bcos, float bsin, const float *p) {
  const float MARGIN = 1e-2f;
  const float angle_cos = bcos, angle_sin = -bsin;
  const float rot_x = (p[0] - box[0]) * angle_cos + (p[1] - box[1]) * (-angle_sin);
  const float rot_y = (p[0] - box[0]) * angle_sin + (p[1] - box[1]) * angle_cos;
  return d6_fabsf(rot_x) < box[3] / 2 + MARGIN && d6_fabsf(rot_y) < box[4] / 2 + MARGIN;
}

/* box_overlap, iou3d_cpu.cpp:128-220 */
D6_HD float d6_box_overlap(const float *box_a, const float *box_b) {
  const float a_angle = box_a[6], b_angle = box_b[6];
  const float a_dx_half = box_a[3] / 2, b_dx_half = box_b[3] / 2;
  const float a_dy_half = box_a[4] / 2, b_dy_half = box_b[4] / 2;
  const float a_x1 = box_a[0] - a_dx_half, a_y1 = box_a[1] - a_dy_half;
  const float a_x2 = box_a[0] + a_dx_half, a_y2 = box_a[1] + a_dy_half;
  const float b_x1 = box_b[0] - b_dx_half, b_y1 = box_b[1] - b_dy_half;
  const float b_x2 = box_b[0] + b_dx_half, b_y2 = box_b[1] + b_dy_half;

  float ca[5][2] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
  float cb[5][2] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};

  float a_sin, a_cos, b_sin, b_cos;
  d6_sincosf(a_angle, &a_sin, &a_cos);
  d6_sincosf(b_angle, &b_sin, &b_cos);

  for (int k = 0; k < 4; ++k) { /* rotate_around_center, iou3d_cpu.cpp:118-122 */
    float nx = (ca[k][0] - box_a[0]) * a_cos + (ca[k][1] - box_a[1]) * (-a_sin) + box_a[0];
    float ny = (ca[k][0] - box_a[0]) * a_sin + (ca[k][1] - box_a[1]) * a_cos + box_a[1];
    ca[k][0] = nx; ca[k][1] = ny;
    nx = (cb[k][0] - box_b[0]) * b_cos + (cb[k][1] - box_b[1]) * (-b_sin) + box_b[0];
    ny = (cb[k][0] - box_b[0]) * b_sin + (cb[k][1] - box_b[1]) * b_cos + box_b[1];
    cb[k][0] = nx; cb[k][1] = ny;
  }
  ca[4][0] = ca[0][0]; ca[4][1] = ca[0][1];
  cb[4][0] = cb[0][0]; cb[4][1] = cb[0][1];

  float cp[16][2];
  float cx = 0.f, cy = 0.f;
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (d6g_seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], cp[cnt])) {
        cx = cx + cp[cnt][0];
        cy = cy + cp[cnt][1];
        ++cnt;
      }
  for (int k = 0; k < 4; ++k) {
    if (d6g_point_in_box2d(box_a, a_cos, a_sin, cb[k])) {
      cx = cx + cb[k][0]; cy = cy + cb[k][1];
      cp[cnt][0] = cb[k][0]; cp[cnt][1] = cb[k][1];
      ++cnt;
    }
    if (d6g_point_in_box2d(box_b, b_cos, b_sin, ca[k])) {
      cx = cx + ca[k][0]; cy = cy + ca[k][1];
      cp[cnt][0] = ca[k][0]; cp[cnt][1] = ca[k][1];
      ++cnt;
    }
  }
  cx /= cnt; /* cnt == 0 -> NaN, harmless: the loops below do not run */
  cy /= cnt;

  /* bubble sort by polar angle about the centroid, iou3d_cpu.cpp:124-126,198-208 */
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i) {
      const float ta = d6_atan2f(cp[i][1] - cy, cp[i][0] - cx);
      const float tb = d6_atan2f(cp[i + 1][1] - cy, cp[i + 1][0] - cx);
      if (ta > tb) {
        const float tx = cp[i][0], ty = cp[i][1];
        cp[i][0] = cp[i + 1][0]; cp[i][1] = cp[i + 1][1];
        cp[i + 1][0] = tx; cp[i + 1][1] = ty;
      }
    }

  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    const float ax = cp[k][0] - cp[0][0], ay = cp[k][1] - cp[0][1];
    const float bx = cp[k + 1][0] - cp[0][0], by = cp[k + 1][1] - cp[0][1];
    area += ax * by - ay * bx;
  }
  return (float)((double)d6_fabsf(area) / 2.0); /* `fabs(area) / 2.0`: double divide */
}

/* iou_bev, iou3d_cpu.cpp:222-229 */
D6_HD float d6_iou_bev(const float *box_a, const float *box_b) {
  const float sa = box_a[3] * box_a[4];
  const float sb = box_b[3] * box_b[4];
  const float s_overlap = d6_box_overlap(box_a, box_b);
  return s_overlap / d6_fmaxf(sa + sb - s_overlap, IOU_EPS);
}

/* iou_normal, iou3d_nms_kernel.cu:314-325 */
D6_HD float d6_iou_normal(const float *a, const float *b) {
  const float left = d6_fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2);
  const float right = d6_fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = d6_fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2);
  const float bottom = d6_fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float width = d6_fmaxf(right - left, 0.f), height = d6_fmaxf(bottom - top, 0.f);
  const float interS = width * height;
  const float Sa = a[3] * a[4];
  const float Sb = b[3] * b[4];
  return interS / d6_fmaxf(Sa + Sb - interS, IOU_EPS);
}


#endif /* DET6D_GEOM_H */
