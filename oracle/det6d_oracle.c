/*
 * det6d_oracle.c — CPU restatement of the Det6D inference hot path of HITSZ-NRSL/De6D.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path (de6d_amd/) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - rotated IoU / NMS: pinned against the reference's own iou3d_cpu.cpp compiled into
 *     oracle/_ref (tests/test_oracle_ref.py) and against fixtures generated from it;
 *   - box decode and the whole-model glue: pinned against golden vectors produced by importing the
 *     reference's Python (tests/golden/make_golden.py);
 *   - FPS / ball query / grouping / three_nn: the reference has NO CPU implementation and no
 *     tests for them, and its CUDA cannot run here: parity unpinned at the reference level.
 *     They are restated line by line from the .cu files cited at each function.
 *
 * Arithmetic conventions shared with the HIP kernels (compile both with -ffp-contract=off):
 *   - squared distances use the contraction LLVM's DAG combiner (and therefore NVVM) produces
 *     for `dx*dx + dy*dy + dz*dz`:  fma(dz,dz, fma(dx,dx, dy*dy));
 *   - rotated-box geometry follows iou3d_cpu.cpp as built for x86-64: no contraction;
 *   - transcendentals come from include/det6d_math.h.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/det6d_math.h"

#define ORACLE_API __attribute__((visibility("default")))

/* The reference writes dx*dx + dy*dy + dz*dz (sampling_gpu.cu:143, ball_query_gpu.cu:39, interpolate_gpu.cu:37) and lets
 * NVCC contract it.  Two contractions are plausible: the one LLVM's DAG combiner emits — fma(dz,dz, fma(dx,dx, dy*dy)), the
 * shipped convention — and SURVEY.md A.2's left-to-right reading fma(dz,dz, fma(dy,dy, dx*dx)).  No NVIDIA toolchain is
 * available to settle it, so the alternative can be switched on (tests/test_contraction_order.py quantifies what it would
 * change; nothing else ever sets it). */
static int g_sqdist_alt = 0;
ORACLE_API void det6d_oracle_set_sqdist_order(int alt) { g_sqdist_alt = alt; }
static inline float sqdist(float dx, float dy, float dz) {
  if (g_sqdist_alt) return D6_FMA(dz, dz, D6_FMA(dy, dy, dx * dx));
  return D6_FMA(dz, dz, D6_FMA(dx, dx, dy * dy));
}

/* core/pcdet/ops/pointnet2/pointnet2_batch/src/cuda_utils.h:10-14 */
ORACLE_API int det6d_oracle_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 1024) t = 1024;
  if (t < 1) t = 1;
  return t;
}

/* sampling_gpu.cu:94-99 (__update) + :159-216 (the halving tree over block_size slots). */
static void tree_reduce(float *dists, int *dists_i, int block_size) {
  for (int s = block_size / 2; s >= 1; s >>= 1) {
    for (int tid = 0; tid < s; ++tid) {
      const float v1 = dists[tid], v2 = dists[tid + s];
      const int i1 = dists_i[tid], i2 = dists_i[tid + s];
      dists[tid] = d6_fmaxf(v1, v2);
      dists_i[tid] = v2 > v1 ? i2 : i1;
    }
  }
}

/* sampling_gpu.cu:101-222 farthest_point_sampling_kernel<block_size>, one block per scene. */
ORACLE_API int det6d_oracle_fps(int b, int n, int m, const float *xyz, float *temp, int *idx) {
  if (b < 0 || n <= 0 || m < 0) return -1;
  if (m == 0) return 0;
  const int S = det6d_oracle_opt_n_threads(n);
#pragma omp parallel for schedule(dynamic)   /* scenes are independent (one CUDA block each) */
  for (int bi = 0; bi < b; ++bi) {
    float *dists = (float *)malloc(sizeof(float) * S);
    int *dists_i = (int *)malloc(sizeof(int) * S);
    const float *ds = xyz + (size_t)bi * n * 3;
    float *tp = temp + (size_t)bi * n;
    int *out = idx + (size_t)bi * m;
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
      for (int tid = 0; tid < S; ++tid) {
        int besti = 0;
        float best = -1.0f;
        for (int k = tid; k < n; k += S) {
          const float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
          const float d = sqdist(x2 - x1, y2 - y1, z2 - z1);
          const float d2 = d6_fminf(d, tp[k]);
          tp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      tree_reduce(dists, dists_i, S);
      old = dists_i[0];
      out[j] = old;
    }
    free(dists);
    free(dists_i);
  }
  return 0;
}

/* sampling_gpu.cu:419-540 furthest_point_sampling_weights_kernel<block_size>. */
ORACLE_API int det6d_oracle_fps_weights(int b, int n, int m, const float *xyz, const float *weights,
                                        float *temp, int *idx) {
  if (b < 0 || n <= 0 || m < 0) return -1;
  if (m == 0) return 0;
  const int S = det6d_oracle_opt_n_threads(n);
#pragma omp parallel for schedule(dynamic)
  for (int bi = 0; bi < b; ++bi) {
    float *dists = (float *)malloc(sizeof(float) * S);
    int *dists_i = (int *)malloc(sizeof(int) * S);
    const float *ds = xyz + (size_t)bi * n * 3;
    const float *w = weights + (size_t)bi * n;
    float *tp = temp + (size_t)bi * n;
    int *out = idx + (size_t)bi * m;
    int old = 0;
    for (int j = 0; j < m; ++j) {
      const float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
      for (int tid = 0; tid < S; ++tid) {
        int besti = 0;
        float best = -1.0f;
        for (int k = tid; k < n; k += S) {
          if (j == 0) {
            const float d = w[k];
            besti = d > best ? k : besti;
            best = d > best ? d : best;
          } else {
            const float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
            float d = sqdist(x2 - x1, y2 - y1, z2 - z1);
            d = d6_fminf(d, tp[k]);
            tp[k] = d;
            /* `d * max(weights[k], 1e-12)`: 1e-12 is a double literal, product formed in
             * double and rounded once to float (sampling_gpu.cu:466). */
            const double wk = fmax((double)w[k], 1e-12);
            const float d2 = (float)((double)d * wk);
            besti = d2 > best ? k : besti;
            best = d2 > best ? d2 : best;
          }
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      tree_reduce(dists, dists_i, S);
      old = dists_i[0];
      out[j] = old;
    }
    free(dists);
    free(dists_i);
  }
  return 0;
}

/* sampling_gpu.cu:16-32 gather_points_kernel_fast */
ORACLE_API int det6d_oracle_gather_points(int b, int c, int n, int npoints, const float *points,
                                          const int *idx, float *out) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int j = 0; j < npoints; ++j)
        out[((size_t)bi * c + ci) * npoints + j] =
            points[((size_t)bi * c + ci) * n + idx[(size_t)bi * npoints + j]];
  return 0;
}

/* sampling_gpu.cu:54-71 gather_points_grad_kernel_fast (sequential sum order on the CPU) */
ORACLE_API int det6d_oracle_gather_points_grad(int b, int c, int n, int npoints,
                                               const float *grad_out, const int *idx,
                                               float *grad_points) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int j = 0; j < npoints; ++j)
        grad_points[((size_t)bi * c + ci) * n + idx[(size_t)bi * npoints + j]] +=
            grad_out[((size_t)bi * c + ci) * npoints + j];
  return 0;
}

/* ball_query_gpu.cu:15-51 ball_query_kernel_fast */
ORACLE_API int det6d_oracle_ball_query(int b, int n, int m, float radius, int nsample,
                                       const float *new_xyz, const float *xyz, int *idx) {
  const float radius2 = radius * radius;
  for (int bi = 0; bi < b; ++bi)
    for (int pi = 0; pi < m; ++pi) {
      const float *q = new_xyz + ((size_t)bi * m + pi) * 3;
      const float *p = xyz + (size_t)bi * n * 3;
      int *o = idx + ((size_t)bi * m + pi) * nsample;
      int cnt = 0;
      for (int k = 0; k < n; ++k) {
        const float d2 = sqdist(q[0] - p[k * 3 + 0], q[1] - p[k * 3 + 1], q[2] - p[k * 3 + 2]);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) o[l] = k;
          o[cnt] = k;
          ++cnt;
          if (cnt >= nsample) break;
        }
      }
    }
  return 0;
}

/* ball_query_gpu.cu:93-130 ball_query_cnt_kernel_fast; :53-91 ball_query_dilated_kernel_fast.
 * radius_in < 0 selects the plain ball (no inner test). */
static int ball_query_shell(int b, int n, int m, int dilated, float radius_in, float radius_out,
                            int nsample, const float *new_xyz, const float *xyz, int *idx_cnt,
                            int *idx) {
  const float rin2 = radius_in * radius_in;
  const float rout2 = radius_out * radius_out;
#pragma omp parallel for collapse(2) schedule(static)   /* one CUDA thread per (scene, centre) */
  for (int bi = 0; bi < b; ++bi)
    for (int pi = 0; pi < m; ++pi) {
      const float *q = new_xyz + ((size_t)bi * m + pi) * 3;
      const float *p = xyz + (size_t)bi * n * 3;
      int *o = idx + ((size_t)bi * m + pi) * nsample;
      int cnt = 0;
      for (int k = 0; k < n; ++k) {
        const float d2 = sqdist(q[0] - p[k * 3 + 0], q[1] - p[k * 3 + 1], q[2] - p[k * 3 + 2]);
        const int hit = dilated ? (d2 >= rin2 && d2 < rout2) : (d2 < rout2);
        if (hit) {
          o[cnt] = k;
          ++cnt;
          if (cnt >= nsample) break;
        }
      }
      idx_cnt[(size_t)bi * m + pi] = cnt;
      for (int l = 0; cnt < nsample; ++l, ++cnt) o[cnt] = o[l];
    }
  return 0;
}
ORACLE_API int det6d_oracle_ball_query_cnt(int b, int n, int m, float radius, int nsample,
                                           const float *new_xyz, const float *xyz, int *idx_cnt,
                                           int *idx) {
  return ball_query_shell(b, n, m, 0, 0.0f, radius, nsample, new_xyz, xyz, idx_cnt, idx);
}
ORACLE_API int det6d_oracle_ball_query_dilated(int b, int n, int m, float radius_in,
                                               float radius_out, int nsample, const float *new_xyz,
                                               const float *xyz, int *idx_cnt, int *idx) {
  return ball_query_shell(b, n, m, 1, radius_in, radius_out, nsample, new_xyz, xyz, idx_cnt, idx);
}

/* fused two-shell form (see include/det6d_ops.h): literally two reference-style queries */
ORACLE_API int det6d_oracle_ball_query_pair(int b, int n, int m, float rin_a, float rout_a, int ns_a,
                                            float rin_b, float rout_b, int ns_b, const float *new_xyz,
                                            const float *xyz, int *cnt_a, int *idx_a, int *cnt_b, int *idx_b) {
  memset(idx_a, 0, sizeof(int) * (size_t)b * m * ns_a);
  memset(idx_b, 0, sizeof(int) * (size_t)b * m * ns_b);
  ball_query_shell(b, n, m, 1, rin_a, rout_a, ns_a, new_xyz, xyz, cnt_a, idx_a);
  return ball_query_shell(b, n, m, 1, rin_b, rout_b, ns_b, new_xyz, xyz, cnt_b, idx_b);
}

ORACLE_API int64_t det6d_oracle_ball_query_grid_workspace_bytes(int b, int n) { (void)b; (void)n; return 0; }
ORACLE_API int det6d_oracle_ball_query_pair_grid(int b, int n, int m, float rin_a, float rout_a, int ns_a,
                                                 float rin_b, float rout_b, int ns_b, const float *new_xyz,
                                                 const float *xyz, void *workspace, int *cnt_a, int *idx_a,
                                                 int *cnt_b, int *idx_b) {
  (void)workspace;   /* the grid is an acceleration structure only: results are the brute-force ones */
  return det6d_oracle_ball_query_pair(b, n, m, rin_a, rout_a, ns_a, rin_b, rout_b, ns_b, new_xyz, xyz, cnt_a, idx_a,
                                      cnt_b, idx_b);
}

/* group_points_gpu.cu:53-72 group_points_kernel_fast */
ORACLE_API int det6d_oracle_group_points(int b, int c, int n, int npoints, int nsample,
                                         const float *points, const int *idx, float *out) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int pi = 0; pi < npoints; ++pi)
        for (int s = 0; s < nsample; ++s)
          out[(((size_t)bi * c + ci) * npoints + pi) * nsample + s] =
              points[((size_t)bi * c + ci) * n + idx[((size_t)bi * npoints + pi) * nsample + s]];
  return 0;
}

/* group_points_gpu.cu:14-31 group_points_grad_kernel_fast */
ORACLE_API int det6d_oracle_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                              const float *grad_out, const int *idx,
                                              float *grad_points) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int pi = 0; pi < npoints; ++pi)
        for (int s = 0; s < nsample; ++s)
          grad_points[((size_t)bi * c + ci) * n + idx[((size_t)bi * npoints + pi) * nsample + s]] +=
              grad_out[(((size_t)bi * c + ci) * npoints + pi) * nsample + s];
  return 0;
}

/* interpolate_gpu.cu:16-59 three_nn_kernel_fast */
ORACLE_API int det6d_oracle_three_nn(int b, int n, int m, const float *unknown, const float *known,
                                     float *dist2, int *idx) {
  for (int bi = 0; bi < b; ++bi)
    for (int pi = 0; pi < n; ++pi) {
      const float *u = unknown + ((size_t)bi * n + pi) * 3;
      const float *kn = known + (size_t)bi * m * 3;
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float d = sqdist(u[0] - kn[k * 3 + 0], u[1] - kn[k * 3 + 1], u[2] - kn[k * 3 + 2]);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      float *od = dist2 + ((size_t)bi * n + pi) * 3;
      int *oi = idx + ((size_t)bi * n + pi) * 3;
      od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
      oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
    }
  return 0;
}

/* interpolate_gpu.cu:84-104 three_interpolate_kernel_fast;
 * `w0*p0 + w1*p1 + w2*p2` contracted as fma(w2,p2, fma(w0,p0, w1*p1)). */
ORACLE_API int det6d_oracle_three_interpolate(int b, int c, int m, int n, const float *points,
                                              const int *idx, const float *weight, float *out) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci) {
      const float *p = points + ((size_t)bi * c + ci) * m;
      for (int pi = 0; pi < n; ++pi) {
        const float *w = weight + ((size_t)bi * n + pi) * 3;
        const int *id = idx + ((size_t)bi * n + pi) * 3;
        out[((size_t)bi * c + ci) * n + pi] =
            D6_FMA(w[2], p[id[2]], D6_FMA(w[0], p[id[0]], w[1] * p[id[1]]));
      }
    }
  return 0;
}

/* interpolate_gpu.cu:127-149 three_interpolate_grad_kernel_fast */
ORACLE_API int det6d_oracle_three_interpolate_grad(int b, int c, int n, int m,
                                                   const float *grad_out, const int *idx,
                                                   const float *weight, float *grad_points) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci) {
      float *gp = grad_points + ((size_t)bi * c + ci) * m;
      for (int pi = 0; pi < n; ++pi) {
        const float g = grad_out[((size_t)bi * c + ci) * n + pi];
        const float *w = weight + ((size_t)bi * n + pi) * 3;
        const int *id = idx + ((size_t)bi * n + pi) * 3;
        gp[id[0]] += g * w[0];
        gp[id[1]] += g * w[1];
        gp[id[2]] += g * w[2];
      }
    }
  return 0;
}

/* Rotated BEV overlap / IoU arithmetic lives in include/det6d_geom.h (shared with the HIP
 * kernels so both sides execute the same fp32 operation sequence); it follows
 * core/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:59-229 and is pinned against that file's own
 * build in oracle/_ref by tests/test_oracle_ref.py. */
#include "../include/det6d_geom.h"
#define box_overlap d6_box_overlap
#define iou_bev d6_iou_bev
#define iou_normal d6_iou_normal

/* boxes_overlap_kernel / boxes_iou_bev_kernel, iou3d_nms_kernel.cu:236-265;
 * boxes_iou_bev_cpu, iou3d_cpu.cpp:232-252 */
ORACLE_API int det6d_oracle_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b,
                                              const float *boxes_b, float *ans) {
  for (int i = 0; i < num_a; ++i)
    for (int j = 0; j < num_b; ++j)
      ans[(size_t)i * num_b + j] = box_overlap(boxes_a + i * 7, boxes_b + j * 7);
  return 0;
}
ORACLE_API int det6d_oracle_boxes_iou_bev(int num_a, const float *boxes_a, int num_b,
                                          const float *boxes_b, float *ans) {
  for (int i = 0; i < num_a; ++i)
    for (int j = 0; j < num_b; ++j)
      ans[(size_t)i * num_b + j] = iou_bev(boxes_a + i * 7, boxes_b + j * 7);
  return 0;
}

ORACLE_API int64_t det6d_oracle_nms_mask_words(int boxes_num) {
  return (int64_t)boxes_num * ((boxes_num + 63) / 64);
}

/* nms_kernel / nms_normal_kernel tiles, iou3d_nms_kernel.cu:267-311,328-372 */
ORACLE_API int det6d_oracle_nms_mask(int boxes_num, const float *boxes, float thresh, int normal,
                                     uint64_t *mask) {
  const int col_blocks = (boxes_num + 63) / 64;
  for (int i = 0; i < boxes_num; ++i) {
    const int row_start = i / 64, t = i % 64;
    for (int col_start = 0; col_start < col_blocks; ++col_start) {
      int col_size = boxes_num - col_start * 64;
      if (col_size > 64) col_size = 64;
      uint64_t bits = 0;
      const int start = (row_start == col_start) ? t + 1 : 0;
      for (int c = start; c < col_size; ++c) {
        const float *bj = boxes + (size_t)(col_start * 64 + c) * 7;
        const float v = normal ? iou_normal(boxes + (size_t)i * 7, bj) : iou_bev(boxes + (size_t)i * 7, bj);
        if (v > thresh) bits |= 1ULL << c;
      }
      mask[(size_t)i * col_blocks + col_start] = bits;
    }
  }
  return 0;
}

/* host greedy scan of nms_gpu, iou3d_nms.cpp:116-132 */
static int nms_greedy(int boxes_num, const uint64_t *mask, int64_t *keep) {
  const int col_blocks = (boxes_num + 63) / 64;
  uint64_t *remv = (uint64_t *)calloc(col_blocks > 0 ? col_blocks : 1, sizeof(uint64_t));
  int num_to_keep = 0;
  for (int i = 0; i < boxes_num; ++i) {
    const int nblock = i / 64, inblock = i % 64;
    if (!(remv[nblock] & (1ULL << inblock))) {
      keep[num_to_keep++] = i;
      const uint64_t *p = mask + (size_t)i * col_blocks;
      for (int j = nblock; j < col_blocks; ++j) remv[j] |= p[j];
    }
  }
  free(remv);
  return num_to_keep;
}

ORACLE_API int det6d_oracle_nms(int boxes_num, const float *boxes, float thresh, uint64_t *mask,
                                int64_t *keep, int *num_keep) {
  det6d_oracle_nms_mask(boxes_num, boxes, thresh, 0, mask);
  *num_keep = nms_greedy(boxes_num, mask, keep);
  return 0;
}
ORACLE_API int det6d_oracle_nms_normal(int boxes_num, const float *boxes, float thresh,
                                       uint64_t *mask, int64_t *keep, int *num_keep) {
  det6d_oracle_nms_mask(boxes_num, boxes, thresh, 1, mask);
  *num_keep = nms_greedy(boxes_num, mask, keep);
  return 0;
}

/* greedy scan over a given IoU matrix (used to derive keep lists from oracle/_ref's matrices) */
ORACLE_API int det6d_oracle_nms_from_iou(int boxes_num, const float *iou, float thresh,
                                         int64_t *keep) {
  const int col_blocks = (boxes_num + 63) / 64;
  uint64_t *mask = (uint64_t *)calloc((size_t)boxes_num * (col_blocks > 0 ? col_blocks : 1) + 1,
                                      sizeof(uint64_t));
  for (int i = 0; i < boxes_num; ++i)
    for (int j = i + 1; j < boxes_num; ++j)
      if (iou[(size_t)i * boxes_num + j] > thresh) mask[(size_t)i * col_blocks + j / 64] |= 1ULL << (j % 64);
  const int n = nms_greedy(boxes_num, mask, keep);
  free(mask);
  return n;
}

/* ------------------------------------------------------------------------------------------
 * Engine-level ops (restating the Python hot loop)
 * ---------------------------------------------------------------------------------------- */

/* PointNet2FSMSG.break_up_pc + view, core/pcdet/models/backbones_3d/pointnet2_backbone.py:193-224 */
ORACLE_API int det6d_oracle_pack_points(int total, int cin, const float *points, int ld,
                                        float *rows, float *xyz_out) {
  for (int i = 0; i < total; ++i) {
    const float *src = points + (size_t)i * (1 + 3 + cin);
    float *dst = rows + (size_t)i * ld;
    for (int c = 0; c < 3 + cin; ++c) dst[c] = src[1 + c];
    for (int c = 3 + cin; c < ld; ++c) dst[c] = 0.f;
    if (xyz_out) for (int c = 0; c < 3; ++c) xyz_out[(size_t)i * 3 + c] = src[1 + c];
  }
  return 0;
}

/* sampler of an SA layer as one call (pointnet2_modules.py:376-450): slice, sigmoid**gamma, FPS, + lo */
ORACLE_API int det6d_oracle_fps_fused(int b, int n_total, int lo, int hi, int m, const float *xyz,
                                      const float *scores, float gamma, float *temp, int *idx,
                                      int idx_stride, int idx_offset) {
  const int n = hi - lo;
  float *sx = (float *)malloc(sizeof(float) * (size_t)n * 3);
  float *sw = (float *)malloc(sizeof(float) * (size_t)n);
  float *st = (float *)malloc(sizeof(float) * (size_t)n);
  int *si = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
  (void)temp;
  for (int bi = 0; bi < b; ++bi) {
    memcpy(sx, xyz + ((size_t)bi * n_total + lo) * 3, sizeof(float) * (size_t)n * 3);
    for (int k = 0; k < n; ++k) st[k] = 1e10f;
    if (scores) {
      for (int k = 0; k < n; ++k) sw[k] = d6_sigmoid_powf(scores[(size_t)bi * n_total + lo + k], gamma);
      det6d_oracle_fps_weights(1, n, m, sx, sw, st, si);
    } else {
      det6d_oracle_fps(1, n, m, sx, st, si);
    }
    for (int j = 0; j < m; ++j) idx[(size_t)bi * idx_stride + idx_offset + j] = si[j] + lo;
  }
  free(sx); free(sw); free(st); free(si);
  return 0;
}

ORACLE_API int det6d_oracle_gather_centres(int b, int n, int m, const float *xyz, const int *idx,
                                           float *xyz_out, float *rows_out, int ld_rows, int zero_from) {
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < m; ++j) {
      const float *src = xyz + ((size_t)bi * n + idx[(size_t)bi * m + j]) * 3;
      for (int c = 0; c < 3; ++c) {
        xyz_out[((size_t)bi * m + j) * 3 + c] = src[c];
        if (rows_out) rows_out[((size_t)bi * m + j) * ld_rows + c] = src[c];
      }
      if (rows_out)
        for (int c = zero_from; c < ld_rows; ++c) rows_out[((size_t)bi * m + j) * ld_rows + c] = 0.f;
    }
  return 0;
}

ORACLE_API int det6d_oracle_with_batch_index(int b, int m, const float *src, int ld_src, int ncol, float *dst) {
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < m; ++j) {
      float *d = dst + ((size_t)bi * m + j) * (ncol + 1);
      d[0] = (float)bi;
      for (int c = 0; c < ncol; ++c) d[1 + c] = src[((size_t)bi * m + j) * ld_src + c];
    }
  return 0;
}

ORACLE_API int det6d_oracle_gather_rows(int b, int n, int m, int ld_in, int ld_out, int ncol,
                                        const float *rows_in, const int *idx, float *rows_out) {
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < m; ++j) {
      const float *src = rows_in + ((size_t)bi * n + idx[(size_t)bi * m + j]) * ld_in;
      float *dst = rows_out + ((size_t)bi * m + j) * ld_out;
      for (int c = 0; c < ncol; ++c) dst[c] = src[c];
    }
  return 0;
}

typedef struct det6d_linear_args {
  int mode, rows, k, ncols;
  const float *a; int lda;
  const float *w; int ldw;
  const float *shift;
  int act;
  float *y; int ldy; int col0;
  int n, m, ns;
  const int *idx;
  const float *ctr; int ldctr;
  int pool;
  const int *cnt;
  const int *hdr, *crow_p, *crow_c;   /* compact row lists (det6d_oracle_compact_groups) */
  int ncols_pad;                      /* zero-filled padding columns (ignored here: numpy callers allocate zeros) */
} det6d_linear_args;

/* Compact row lists: plain sequential restatement of de6d_amd/csrc/compact.hip (this is a data structure of the
 * BUILD, not of the reference: the reference evaluates all nsample rows, ball_query_gpu.cu:75-90 pads them with
 * repetitions of the first cnt hits; the parity tests compare the compact path with this oracle's DENSE rows). */
ORACLE_API int det6d_oracle_compact_rows_capacity(int total_centres, int ns) {
  return (total_centres * ns + 6 * 128 + 1023) & ~1023;
}
ORACLE_API int det6d_oracle_compact_hdr_ints(int total_centres) { return 16 + 7 * ((total_centres + 255) / 256 + 1); }
static int compact_rows_of(int cnt, int ns, int smin, int split) {
  const int k = cnt < 1 ? 1 : (cnt > ns ? ns : cnt);
  if (split > 0 && k > split) return (k + split - 1) / split * split;
  int rows = smin;
  while (rows < k) rows <<= 1;
  return rows;
}
ORACLE_API int det6d_oracle_compact_groups(int b, int n, int m, int ns, int smin, int split, const int *cnt,
                                           const int *idx, int *hdr, int *crow_p, int *crow_c, float *zero_y, int ldy,
                                           int col0, int width) {
  const int total = b * m;
  if (zero_y)
    for (int i = 0; i < total; ++i)
      for (int c = 0; c < width; ++c) zero_y[(size_t)i * ldy + col0 + c] = 0.f;
  if (split > ns) split = ns;
  int count[6] = {0, 0, 0, 0, 0, 0}, start[7], next[6], real = 0, unaligned = 0;
  for (int i = 0; i < total; ++i) {
    const int rows = compact_rows_of(cnt[i], ns, smin, split);
    for (int c = 0; c < 6; ++c) count[c] += (rows & (32 >> c)) != 0;
    real += cnt[i] < ns ? (cnt[i] < 0 ? 0 : cnt[i]) : ns;
  }
  int r = 0;
  for (int c = 0; c < 6; ++c) {
    start[c] = next[c] = r;
    unaligned += count[c] * (32 >> c);
    r = (r + count[c] * (32 >> c) + 127) & ~127;
  }
  start[6] = r;
  hdr[0] = r;
  for (int c = 0; c < 6; ++c) hdr[1 + c] = start[c + 1];
  hdr[7] = total; hdr[8] = real; hdr[9] = unaligned;
  hdr[10] = hdr[11] = 0;   /* tile ticket / exit counter of the HIP group kernels: zero whenever no kernel runs */
  for (int c = 0; c < 6; ++c)
    for (int q = start[c] + count[c] * (32 >> c); q < start[c + 1]; ++q) { crow_p[q] = 0; crow_c[q] = -1; }
  for (int i = 0; i < total; ++i) {
    const int rows = compact_rows_of(cnt[i], ns, smin, split);
    int tag = i;
    if (cnt[i] <= 0) tag |= 0x40000000;
    if (rows & (rows - 1)) tag |= 0x20000000;
    for (int c = 0; c < 6; ++c) {
      const int sz = 32 >> c;
      if (!(rows & sz)) continue;
      const int off = rows & ~(2 * sz - 1);
      for (int t = 0; t < sz; ++t) {
        crow_p[next[c] + t] = (i / m) * n + idx[(size_t)i * ns + off + t];
        crow_c[next[c] + t] = tag;
      }
      next[c] += sz;
    }
  }
  return 0;
}

ORACLE_API int det6d_oracle_compact_groups_pair(int b, int n, int m, int smin, int split, int ns_a, const int *cnt_a,
                                                const int *idx_a, int *hdr_a, int *crow_p_a, int *crow_c_a, int col0_a,
                                                int width_a, int ns_b, const int *cnt_b, const int *idx_b, int *hdr_b,
                                                int *crow_p_b, int *crow_c_b, int col0_b, int width_b, float *zero_y,
                                                int ldy) {
  const int sm_a = smin < ns_a ? smin : ns_a, sm_b = smin < ns_b ? smin : ns_b;
  det6d_oracle_compact_groups(b, n, m, ns_a, sm_a, split && split < sm_a ? sm_a : split, cnt_a, idx_a, hdr_a, crow_p_a, crow_c_a,
                              zero_y, ldy, col0_a, width_a);
  return det6d_oracle_compact_groups(b, n, m, ns_b, sm_b, split && split < sm_b ? sm_b : split, cnt_b, idx_b, hdr_b, crow_p_b,
                                     crow_c_b, zero_y, ldy, col0_b, width_b);
}

/* The engine pair det6d_ball_query_pair_grid_lists + det6d_compact_groups_pair_counted (csrc/ball_query_grid.hip, compact.hip):
 * the query's hits and counts are the brute-force ones; the contract for the index rows is "slots below the next power of
 * two >= max(cnt, 4) hold the reference's (cyclically padded) row, the rest is untouched" — the restatement writes exactly
 * those slots; the per-block part counts go to hdr + 16 as the HIP query leaves them (7 ints per block of 256 centres). */
ORACLE_API int det6d_oracle_ball_query_pair_grid_lists(int b, int n, int m, float rin_a, float rout_a, int ns_a, float rin_b,
                                                       float rout_b, int ns_b, const float *new_xyz, const float *xyz,
                                                       void *workspace, int *cnt_a, int *idx_a, int *cnt_b, int *idx_b,
                                                       int smin, int split, int *hdr_a, int *hdr_b) {
  (void)workspace;
  const int total = b * m;
  int *full_a = (int *)malloc(sizeof(int) * (size_t)total * ns_a), *full_b = (int *)malloc(sizeof(int) * (size_t)total * ns_b);
  det6d_oracle_ball_query_pair(b, n, m, rin_a, rout_a, ns_a, rin_b, rout_b, ns_b, new_xyz, xyz, cnt_a, full_a, cnt_b, full_b);
  for (int g = 0; g < 2; ++g) {
    const int ns = g ? ns_b : ns_a, *cnt = g ? cnt_b : cnt_a, *full = g ? full_b : full_a;
    int *idx = g ? idx_b : idx_a, *table = (g ? hdr_b : hdr_a) + 16;
    const int sm = smin < ns ? smin : ns;
    int sp = split && split < sm ? sm : split;
    if (sp > ns) sp = ns;
    for (int i = 0; i < total; ++i) {
      int need = 4;
      while (need < cnt[i]) need <<= 1;
      for (int l = 0; l < ns && l < need; ++l) idx[(size_t)i * ns + l] = full[(size_t)i * ns + l];
    }
    for (int bk = 0; bk * 256 < total; ++bk) {
      int *t = table + bk * 7;
      for (int c = 0; c < 7; ++c) t[c] = 0;
      for (int i = bk * 256; i < total && i < bk * 256 + 256; ++i) {
        const int rows = compact_rows_of(cnt[i], ns, sm, sp);
        for (int c = 0; c < 6; ++c) t[c] += (rows & (32 >> c)) != 0;
        t[6] += cnt[i] < ns ? (cnt[i] < 0 ? 0 : cnt[i]) : ns;
      }
    }
  }
  free(full_a); free(full_b);
  return 0;
}
ORACLE_API int det6d_oracle_compact_groups_pair_counted(int b, int n, int m, int smin, int split, int ns_a, const int *cnt_a,
                                                        const int *idx_a, int *hdr_a, int *crow_p_a, int *crow_c_a, int col0_a,
                                                        int width_a, int ns_b, const int *cnt_b, const int *idx_b, int *hdr_b,
                                                        int *crow_p_b, int *crow_c_b, int col0_b, int width_b, float *zero_y,
                                                        int ldy) {
  /* (the sequential builder counts for itself: the tables only save the parallel builder a launch) */
  return det6d_oracle_compact_groups_pair(b, n, m, smin, split, ns_a, cnt_a, idx_a, hdr_a, crow_p_a, crow_c_a, col0_a, width_a, ns_b,
                                          cnt_b, idx_b, hdr_b, crow_p_b, crow_c_b, col0_b, width_b, zero_y, ldy);
}

/* Order of the fma chain of one output.  Plain rows: k ascending.  GATHERED rows [x - cx, y - cy, z - cz, f_0 ..] (the first
 * layer of a grouped MLP): the feature columns first (k = 3 .. K-1), the three relative coordinates LAST.  The reference's
 * Conv2d over torch.cat([grouped_xyz, grouped_features]) (pointnet2_utils.py:449-455, pointnet2_modules.py:561-568) is a
 * cuDNN / ATen GEMM whose summation order is unspecified, so any fixed order restates it; this one makes the feature part
 * of the chain a function of the POINT alone, which the HIP path computes once per point instead of once per
 * (centre, neighbour) row (csrc/expand.hip).  Pinned like every other order: tests/golden/det6d_tiny.npz, 1e-4 abs. */
static inline int chain_k(int kk, int K, int gathered) {
  if (!gathered || K <= 3) return kk;
  return kk + 3 < K ? kk + 3 : kk + 3 - K;
}

/* det6d_linear over a compact row list: rows = hdr[0]; mode 2 gathers through crow_p / crow_c; pool = -1 takes the
 * maximum over the rows of every centre (empty balls -> 0): a centre in ONE part overwrites y, a centre cut into
 * several parts is max-combined with what y holds (the caller zeroes it), like the kernels' atomic max. */
static int oracle_linear_compact(const det6d_linear_args *g) {
  const int K = g->k, N = g->ncols, R = g->hdr[0], centres = g->hdr[7];
  float *arow = (float *)malloc(sizeof(float) * (K > 0 ? K : 1));
  float *acc = (float *)malloc(sizeof(float) * N);
  char *seen = (char *)calloc((size_t)centres + 1, 1);
  for (int r = 0; r < R; ++r) {
    const int tag = (g->mode == 2 || g->pool < 0) ? g->crow_c[r] : 0;
    if (tag < 0) continue;   /* alignment row */
    const int cj = tag & 0x1fffffff;
    if (g->mode == 2) {
      const float *src = g->a + (size_t)g->crow_p[r] * g->lda;
      const float *c = g->ctr + (size_t)cj * g->ldctr;
      for (int k = 0; k < K; ++k) arow[k] = k < 3 ? src[k] - c[k] : src[k];
    } else {
      const float *src = g->a + (size_t)r * g->lda;
      for (int k = 0; k < K; ++k) arow[k] = src[k];
    }
    for (int c = 0; c < N; ++c) acc[c] = 0.f;
    for (int kk = 0; kk < K; ++kk) {
      const int k = chain_k(kk, K, g->mode == 2);
      const float av = arow[k];
      const float *wr = g->w + (size_t)k * g->ldw;
      for (int c = 0; c < N; ++c) acc[c] = D6_FMA(av, wr[c], acc[c]);
    }
    for (int c = 0; c < N; ++c) {
      float v = g->shift ? acc[c] + g->shift[c] : acc[c];
      if (g->act == 1) v = v > 0.f ? v : 0.f;
      acc[c] = v;
    }
    if (g->pool < 0) {
      float *dst = g->y + (size_t)cj * g->ldy + g->col0;
      const int empty = (tag & 0x40000000) != 0, multi = (tag & 0x20000000) != 0;
      for (int c = 0; c < N; ++c) {
        const float v = empty ? 0.f : acc[c];
        dst[c] = (!multi && !seen[cj]) ? v : (v > dst[c] ? v : dst[c]);
      }
      seen[cj] = 1;
    } else {
      float *dst = g->y + (size_t)r * g->ldy + g->col0;
      for (int c = 0; c < N; ++c) dst[c] = acc[c];
    }
  }
  free(arow); free(acc); free(seen);
  return 0;
}

/* Conv(1x1, bias=False) -> BatchNorm(eval) -> ReLU with BN folded into W/shift
 * (pointnet2_modules.py:561-568), grouping (pointnet2_utils.py:449-455), mask + max-pool
 * (pointnet2_modules.py:465-472).  Every output is ONE ascending-k fmaf chain starting at 0,
 * then `+ shift`, then ReLU — the exact arithmetic of v_mfma_f32_32x32x2_f32. */
ORACLE_API int det6d_oracle_linear(const det6d_linear_args *g) {
  const int K = g->k, N = g->ncols;
  if (g->hdr) return oracle_linear_compact(g);
  if (g->pool && (g->rows % g->pool)) return -1;
#pragma omp parallel
  {
    float *arow = (float *)malloc(sizeof(float) * (K > 0 ? K : 1));
    float *acc = (float *)malloc(sizeof(float) * N);
    float *best = (float *)malloc(sizeof(float) * N);
    const int ngroups = g->pool ? g->rows / g->pool : g->rows;
    const int per = g->pool ? g->pool : 1;
#pragma omp for schedule(static)
    for (int gi = 0; gi < ngroups; ++gi) {
      for (int s = 0; s < per; ++s) {
        const int r = gi * per + s;
        if (g->mode == 1) {
          const int cj = r / g->ns;          /* flat centre index b*m + j */
          const int bi = cj / g->m;
          const int p = g->idx[r];
          const float *src = g->a + ((size_t)bi * g->n + p) * g->lda;
          const float *c = g->ctr + (size_t)cj * g->ldctr;
          for (int k = 0; k < K; ++k) arow[k] = k < 3 ? src[k] - c[k] : src[k];
        } else {
          const float *src = g->a + (size_t)r * g->lda;
          for (int k = 0; k < K; ++k) arow[k] = src[k];
        }
        for (int c = 0; c < N; ++c) acc[c] = 0.f;
        for (int kk = 0; kk < K; ++kk) {
          const int k = chain_k(kk, K, g->mode == 1);
          const float av = arow[k];
          const float *wr = g->w + (size_t)k * g->ldw;
          for (int c = 0; c < N; ++c) acc[c] = D6_FMA(av, wr[c], acc[c]);
        }
        for (int c = 0; c < N; ++c) {
          float v = g->shift ? acc[c] + g->shift[c] : acc[c];
          if (g->act == 1) v = v > 0.f ? v : 0.f;
          acc[c] = v;
        }
        if (g->pool) {
          if (s == 0) for (int c = 0; c < N; ++c) best[c] = acc[c];
          else for (int c = 0; c < N; ++c) best[c] = acc[c] > best[c] ? acc[c] : best[c];
        } else {
          float *dst = g->y + (size_t)r * g->ldy + g->col0;
          for (int c = 0; c < N; ++c) dst[c] = acc[c];
        }
      }
      if (g->pool) {
        const int live = g->cnt ? g->cnt[gi] > 0 : 1;
        float *dst = g->y + (size_t)gi * g->ldy + g->col0;
        for (int c = 0; c < N; ++c) dst[c] = live ? best[c] : 0.f;
      }
    }
    free(arow); free(acc); free(best);
  }
  return 0;
}

/* det6d_group_expand (csrc/expand.hip), restated LITERALLY: the first layer of a grouped MLP from the per-point partial
 * sums P (the chain over the feature columns, one det6d_oracle_linear over the points with the coordinate rows of W zeroed):
 *   out[r][c] = act(fma(dz, W[2][c], fma(dy, W[1][c], fma(dx, W[0][c], P[p(r)][pcol0 + c]))) + shift[c]), [c1, ldo) zero.
 * tests/test_oracle_props.py checks on the CPU that this IS det6d_oracle_linear in the gathered modes (chain_k order). */
ORACLE_API int det6d_oracle_group_expand(int rows, int c1, const float *p, int ldp, int pcol0, const float *w, int ldw,
                                         const float *shift, int act, const float *pts, int ldpts, const float *ctr, int ldctr,
                                         const int *idx, int n, int m, int ns, const int *hdr, const int *crow_p,
                                         const int *crow_c, float *out, int ldo) {
  const int live = hdr ? hdr[0] : rows;
  for (int r = 0; r < live; ++r) {
    float *o = out + (size_t)r * ldo;
    for (int c = 0; c < ldo; ++c) o[c] = 0.f;
    long long prow;
    int cj;
    if (hdr) {
      const int tag = crow_c[r];
      if (tag < 0) continue;                       /* alignment row */
      cj = tag & 0x1fffffff;
      prow = crow_p[r];
    } else {
      cj = r / ns;
      prow = (long long)(cj / m) * n + idx[r];
    }
    const float *pt = pts + prow * ldpts, *ce = ctr + (size_t)cj * ldctr;
    const float dx = pt[0] - ce[0], dy = pt[1] - ce[1], dz = pt[2] - ce[2];
    for (int c = 0; c < c1; ++c) {
      float v = D6_FMA(dz, w[2 * (size_t)ldw + c], D6_FMA(dy, w[ldw + c], D6_FMA(dx, w[c], p[prow * ldp + pcol0 + c])));
      v = shift ? v + shift[c] : v;
      if (act == 1) v = v > 0.f ? v : 0.f;
      o[c] = v;
    }
  }
  return 0;
}

/* det6d_mlp_group3 (csrc/mlp_group.hip): by definition expand -> linear -> linear(pool) */
ORACLE_API int det6d_oracle_mlp_group3(int rows, const float *p, int ldp, int pcol0, const float *w1, int ldw1, const float *s1, int c1,
                                       const float *w2, int ldw2, const float *s2, int c2, const float *w3, int ldw3, const float *s3,
                                       int c3, const float *pts, int ldpts, const float *ctr, int ldctr, const int *idx, int n, int m,
                                       int ns, const int *cnt, const int *hdr, const int *crow_p, const int *crow_c, float *y, int ldy,
                                       int col0) {
  float *h1 = (float *)calloc((size_t)rows * c1 + 1, sizeof(float));
  float *h2 = (float *)calloc((size_t)rows * c2 + 1, sizeof(float));
  det6d_oracle_group_expand(rows, c1, p, ldp, pcol0, w1, ldw1, s1, 1, pts, ldpts, ctr, ldctr, idx, n, m, ns, hdr, crow_p, crow_c, h1, c1);
  det6d_linear_args g;
  memset(&g, 0, sizeof(g));
  g.mode = 0; g.rows = rows; g.k = c1; g.ncols = c2; g.a = h1; g.lda = c1; g.w = w2; g.ldw = ldw2; g.shift = s2;
  g.act = 1; g.y = h2; g.ldy = c2; g.hdr = hdr;
  det6d_oracle_linear(&g);
  memset(&g, 0, sizeof(g));
  g.mode = 0; g.rows = rows; g.k = c2; g.ncols = c3; g.a = h2; g.lda = c2; g.w = w3; g.ldw = ldw3; g.shift = s3;
  g.act = 1; g.y = y; g.ldy = ldy; g.col0 = col0;
  if (hdr) { g.pool = -1; g.hdr = hdr; g.crow_c = crow_c; } else { g.pool = ns; g.cnt = cnt; }
  const int rc = det6d_oracle_linear(&g);
  free(h1); free(h2);
  return rc;
}

/* det6d_mlp_rows (csrc/mlp_rows.hip): by definition one det6d_oracle_linear per layer */
typedef struct det6d_rows_layer {
  const float *w; int ldw; int wrow0;
  const float *shift;
  int k, n, act;
  float *out; int ldo; int ocol0;
} det6d_rows_layer;
ORACLE_API int det6d_oracle_mlp_rows(int rows, const float *x, int ldx, int xcol0, int nchains, const int *nlayers,
                                     const det6d_rows_layer *layers) {
  int off = 0;
  for (int c = 0; c < nchains; ++c) {
    const float *in = x + xcol0;
    int ldin = ldx;
    float *prev = NULL;
    for (int l = 0; l < nlayers[c]; ++l) {
      const det6d_rows_layer *L = &layers[off + l];
      float *h = (float *)calloc((size_t)rows * L->n + 1, sizeof(float));
      det6d_linear_args g;
      memset(&g, 0, sizeof(g));
      g.mode = 0; g.rows = rows; g.k = L->k; g.ncols = L->n; g.a = in; g.lda = ldin; g.w = L->w + (size_t)L->wrow0 * L->ldw;
      g.ldw = L->ldw; g.shift = L->shift; g.act = L->act; g.y = h; g.ldy = L->n;
      det6d_oracle_linear(&g);
      if (L->out)
        for (int r = 0; r < rows; ++r)
          for (int j = 0; j < L->n; ++j) L->out[(size_t)r * L->ldo + L->ocol0 + j] = h[(size_t)r * L->n + j];
      free(prev);
      prev = h; in = h; ldin = L->n;
    }
    free(prev);
    off += nlayers[c];
  }
  return 0;
}

/* fused narrow-MLP entry: by definition the three-call sequence */
ORACLE_API int det6d_oracle_mlp_chain3(int rows, int n, int m, int ns, const float *a, int lda, const int *idx,
                                       const float *ctr, int ldctr, const int *cnt, const float *w1, int ldw1,
                                       const float *s1, int c1, const float *w2, int ldw2, const float *s2, int c2,
                                       const float *w3, int ldw3, const float *s3, int c3, float *y, int ldy,
                                       int col0) {
  float *h1 = (float *)calloc((size_t)rows * c1 + 1, sizeof(float));
  float *h2 = (float *)calloc((size_t)rows * c2 + 1, sizeof(float));
  det6d_linear_args g;
  memset(&g, 0, sizeof(g));
  g.mode = 1; g.rows = rows; g.k = lda; g.ncols = c1; g.a = a; g.lda = lda; g.w = w1; g.ldw = ldw1; g.shift = s1;
  g.act = 1; g.y = h1; g.ldy = c1; g.n = n; g.m = m; g.ns = ns; g.idx = idx; g.ctr = ctr; g.ldctr = ldctr;
  det6d_oracle_linear(&g);
  memset(&g, 0, sizeof(g));
  g.mode = 0; g.rows = rows; g.k = c1; g.ncols = c2; g.a = h1; g.lda = c1; g.w = w2; g.ldw = ldw2; g.shift = s2;
  g.act = 1; g.y = h2; g.ldy = c2;
  det6d_oracle_linear(&g);
  memset(&g, 0, sizeof(g));
  g.mode = 0; g.rows = rows; g.k = c2; g.ncols = c3; g.a = h2; g.lda = c2; g.w = w3; g.ldw = ldw3; g.shift = s3;
  g.act = 1; g.y = y; g.ldy = ldy; g.col0 = col0; g.pool = ns; g.cnt = cnt;
  const int rc = det6d_oracle_linear(&g);
  free(h1); free(h2);
  return rc;
}

/* the same chain over a compact row list: by definition the three-call sequence */
ORACLE_API int det6d_oracle_mlp_chain3_compact(int capacity, const int *hdr, const int *crow_p, const int *crow_c,
                                               const float *a, int lda, const float *ctr, int ldctr, const float *w1,
                                               int ldw1, const float *s1, int c1, const float *w2, int ldw2,
                                               const float *s2, int c2, const float *w3, int ldw3, const float *s3,
                                               int c3, float *y, int ldy, int col0) {
  float *h1 = (float *)calloc((size_t)capacity * c1 + 1, sizeof(float));
  float *h2 = (float *)calloc((size_t)capacity * c2 + 1, sizeof(float));
  det6d_linear_args g;
  memset(&g, 0, sizeof(g));
  g.mode = 2; g.rows = capacity; g.k = lda; g.ncols = c1; g.a = a; g.lda = lda; g.w = w1; g.ldw = ldw1; g.shift = s1;
  g.act = 1; g.y = h1; g.ldy = c1; g.ctr = ctr; g.ldctr = ldctr; g.hdr = hdr; g.crow_p = crow_p; g.crow_c = crow_c;
  det6d_oracle_linear(&g);
  memset(&g, 0, sizeof(g));
  g.mode = 0; g.rows = capacity; g.k = c1; g.ncols = c2; g.a = h1; g.lda = c1; g.w = w2; g.ldw = ldw2; g.shift = s2;
  g.act = 1; g.y = h2; g.ldy = c2; g.hdr = hdr;
  det6d_oracle_linear(&g);
  memset(&g, 0, sizeof(g));
  g.mode = 0; g.rows = capacity; g.k = c2; g.ncols = c3; g.a = h2; g.lda = c2; g.w = w3; g.ldw = ldw3; g.shift = s3;
  g.act = 1; g.y = y; g.ldy = ldy; g.col0 = col0; g.pool = -1; g.hdr = hdr; g.crow_c = crow_c;
  const int rc = det6d_oracle_linear(&g);
  free(h1); free(h2);
  return rc;
}

/* pointnet2_modules.py:465-472: new_features *= (idx_cnt > 0); F.max_pool2d(kernel_size=[1, nsample]) — any nsample */
ORACLE_API int det6d_oracle_group_maxpool(int groups, int ns, int ncols, const float *x, int ldx, const int *cnt, float *y,
                                          int ldy, int col0) {
  if (groups < 0 || ns <= 0 || ncols <= 0) return -1;
  for (int r = 0; r < groups; ++r)
    for (int c = 0; c < ncols; ++c) {
      const float mask = (!cnt || cnt[r] > 0) ? 1.0f : 0.0f;
      float v = x[((size_t)r * ns) * ldx + c] * mask;
      for (int s = 1; s < ns; ++s) v = d6_fmaxf(v, x[((size_t)r * ns + s) * ldx + c] * mask);
      y[(size_t)r * ldy + col0 + c] = v;
    }
  return 0;
}

/* pointnet2_modules.py:415-419 */
ORACLE_API int det6d_oracle_sigmoid_pow(int count, const float *scores, float gamma, float *weights) {
  for (int i = 0; i < count; ++i) weights[i] = d6_sigmoid_powf(scores[i], gamma);
  return 0;
}

/* point_head_box6d_vote.py:816-821: torch.max(off, -R) then torch.min(., R), then add */
ORACLE_API int det6d_oracle_vote_points(int rows, const float *off, int ldo, const float *cand,
                                        int ldc, float rx, float ry, float rz, float *vote, int ldv,
                                        float *off_out) {
  const float R[3] = {rx, ry, rz};
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < 3; ++c) {
      float o = off[(size_t)r * ldo + c];
      o = o > -R[c] ? o : -R[c];
      o = o < R[c] ? o : R[c];
      if (off_out) off_out[(size_t)r * 3 + c] = o;
      vote[(size_t)r * ldv + c] = cand[(size_t)r * ldc + c] + o;
    }
  return 0;
}

/* PointBinResidual6DCoder.decode_torch (use_mean_size=False),
 * core/pcdet/utils/box_coder_utils.py:589-603 (yaw bins), :622-640 (pitch), :642-680 (kernel) */
ORACLE_API int det6d_oracle_decode_boxes(int rows, int nbin, int ground_aware, int minus,
                                         float threshold_rad, float factor_rad, const float *code,
                                         int ldcode, const float *pts, int ldp, float *boxes) {
  const float per_bin = (float)(3.14159265358979323846 * 2.0 / (double)nbin);
  for (int r = 0; r < rows; ++r) {
    const float *c = code + (size_t)r * ldcode;
    const float *p = pts + (size_t)r * ldp;
    float *o = boxes + (size_t)r * 9;
    o[0] = c[0] + p[0];
    o[1] = c[1] + p[1];
    o[2] = c[2] + p[2];
    o[3] = d6_expf(c[3]);
    o[4] = d6_expf(c[4]);
    o[5] = d6_expf(c[5]);
    const float *bin = c + 6, *res = c + 6 + nbin, *gr = c + 6 + 2 * nbin;
    int am = 0;
    for (int i = 1; i < nbin; ++i)
      if (bin[i] > bin[am]) am = i;
    o[6] = ((float)am + res[am]) * per_bin;
    if (ground_aware) {
      const int no_pitch = d6_sigmoidf(gr[0]) < 0.5f;
      float pitch = minus ? gr[1] * factor_rad : (-threshold_rad) - gr[1] * factor_rad;
      if (no_pitch) pitch = 0.f;
      o[7] = pitch;
    } else {
      o[7] = gr[0];
    }
    o[8] = 0.f;
  }
  return 0;
}

/* Detector3DTemplate.post_processing (eval, class-agnostic NMS),
 * core/pcdet/models/detectors/detector3d_template.py:178-284,
 * core/pcdet/models/model_utils/model_nms_utils.py:6-25,
 * core/pcdet/ops/iou3d_nms/iou3d_nms_utils.py:84-99 */
ORACLE_API int det6d_oracle_postprocess(int b, int p, int ncls, const float *cls,
                                        const float *boxes, float score_thr, int pre_max,
                                        int post_max, float nms_thr, float *out_boxes,
                                        float *out_scores, int *out_labels, int *out_index,
                                        int *out_count) {
  float *score = (float *)malloc(sizeof(float) * p);
  int *label = (int *)malloc(sizeof(int) * p);
  int *order = (int *)malloc(sizeof(int) * p);
  float *sorted = (float *)malloc(sizeof(float) * 7 * p);
  uint64_t *mask = (uint64_t *)malloc(sizeof(uint64_t) * (det6d_oracle_nms_mask_words(p) + 1));
  int64_t *keep = (int64_t *)malloc(sizeof(int64_t) * p);
  for (int bi = 0; bi < b; ++bi) {
    int cand = 0;
    for (int i = 0; i < p; ++i) {
      const float *lg = cls + ((size_t)bi * p + i) * ncls;
      float best = d6_sigmoidf(lg[0]);
      int bl = 0;
      for (int c = 1; c < ncls; ++c) {
        const float s = d6_sigmoidf(lg[c]);
        if (s > best) { best = s; bl = c; }
      }
      score[i] = best;
      label[i] = bl + 1;
      if (best >= score_thr) order[cand++] = i;
    }
    /* stable descending insertion sort == topk(sorted) + sort(descending) with ties by index */
    for (int i = 1; i < cand; ++i) {
      const int v = order[i];
      int j = i - 1;
      while (j >= 0 && score[order[j]] < score[v]) { order[j + 1] = order[j]; --j; }
      order[j + 1] = v;
    }
    if (cand > pre_max) cand = pre_max;
    for (int i = 0; i < cand; ++i)
      for (int c = 0; c < 7; ++c) sorted[i * 7 + c] = boxes[((size_t)bi * p + order[i]) * 9 + c];
    int nkeep = 0;
    if (cand > 0) det6d_oracle_nms(cand, sorted, nms_thr, mask, keep, &nkeep);
    if (nkeep > post_max) nkeep = post_max;
    out_count[bi] = nkeep;
    for (int i = 0; i < post_max; ++i) {
      float *ob = out_boxes + ((size_t)bi * post_max + i) * 9;
      if (i < nkeep) {
        const int src = order[keep[i]];
        for (int c = 0; c < 9; ++c) ob[c] = boxes[((size_t)bi * p + src) * 9 + c];
        out_scores[(size_t)bi * post_max + i] = score[src];
        out_labels[(size_t)bi * post_max + i] = label[src];
        out_index[(size_t)bi * post_max + i] = src;
      } else {
        for (int c = 0; c < 9; ++c) ob[c] = 0.f;
        out_scores[(size_t)bi * post_max + i] = 0.f;
        out_labels[(size_t)bi * post_max + i] = 0;
        out_index[(size_t)bi * post_max + i] = -1;
      }
    }
  }
  free(score); free(label); free(order); free(sorted); free(mask); free(keep);
  return 0;
}

ORACLE_API int64_t det6d_oracle_postprocess_workspace_bytes(int b) { (void)b; return 0; }

/* math probes for tests/test_math.py */
ORACLE_API void det6d_oracle_math(int fn, int count, const float *x, const float *y, float *out) {
  for (int i = 0; i < count; ++i) {
    switch (fn) {
      case 0: out[i] = d6_expf(x[i]); break;
      case 1: out[i] = d6_logf(x[i]); break;
      case 2: out[i] = d6_sinf(x[i]); break;
      case 3: out[i] = d6_cosf(x[i]); break;
      case 4: out[i] = d6_atan2f(y[i], x[i]); break;
      case 5: out[i] = d6_sigmoidf(x[i]); break;
      default: out[i] = 0.f;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Input producer: range mask + sample_points + collate 'points' + batch index.
 *   mask_points_by_range   core/pcdet/utils/common_utils.py:61-64 (x and y only)
 *   sample_points          core/pcdet/datasets/processor/data_processor.py:145-178
 *   collate_batch          core/pcdet/datasets/dataset.py:171-176
 * Written as the reference writes it — build the in-range list, then the `choice` list branch by
 * branch, then shuffle — with numpy's generator replaced by the keyed bijections of
 * include/det6d_rng.h:  np.random.choice(S, k, replace=False) := {S[r] : perm_|S|(r) < k} placed at
 * perm(r);  np.random.choice(S, k, replace=True)[e] := S[randint(e)];  np.random.shuffle :=
 * slot -> perm_N(slot).
 * ---------------------------------------------------------------------------------------- */
#include "../include/det6d_rng.h"

ORACLE_API int det6d_oracle_prepare_points(int b, const int *raw_offsets, const int *scene_ids, int c,
                                           const float *raw,
                                           float x_min, float y_min, float x_max, float y_max,
                                           int num_points, float near_depth, uint64_t seed,
                                           float *points_out, int *n_in_range) {
  const int N = num_points;
  for (int s = 0; s < b; ++s) {
    const int lo = raw_offsets[s], n_raw = raw_offsets[s + 1] - lo;
    int *in_idx = (int *)malloc(sizeof(int) * (size_t)(n_raw > 0 ? n_raw : 1));
    char *is_near = (char *)malloc((size_t)(n_raw > 0 ? n_raw : 1));
    int *choice = (int *)malloc(sizeof(int) * (size_t)N);
    int n_in = 0, n_near = 0;
    for (int i = 0; i < n_raw; ++i) {
      const float *p = raw + (size_t)(lo + i) * c;
      if (p[0] >= x_min && p[0] <= x_max && p[1] >= y_min && p[1] <= y_max) {
        const float depth = sqrtf((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);  /* np.linalg.norm, float32 */
        is_near[n_in] = depth < near_depth;
        n_near += is_near[n_in];
        in_idx[n_in++] = i;
      }
    }
    n_in_range[s] = n_in;
    float *out = points_out + (size_t)s * N * (1 + c);
    if (n_in == 0) {
      for (int k = 0; k < N; ++k) {
        out[(size_t)k * (1 + c)] = (float)s;
        for (int j = 0; j < c; ++j) out[(size_t)k * (1 + c) + 1 + j] = 0.f;
      }
      free(in_idx); free(is_near); free(choice);
      continue;
    }
    const uint32_t sid = scene_ids ? (uint32_t)scene_ids[s] : (uint32_t)s;
    const uint32_t key_sel = d6_stream_key(seed, sid, 1);
    const uint32_t key_shuffle = d6_stream_key(seed, sid, 2);
    const uint32_t key_extra = d6_stream_key(seed, sid, 3);
    const int n_far = n_in - n_near;
    if (N < n_in) {
      if (N > n_far) {  /* near_idxs_choice = choice(near, N - n_far, replace=False); concat(near_choice, far) */
        const int k = N - n_far;
        int r_near = 0, r_far = 0;
        for (int r = 0; r < n_in; ++r) {
          if (is_near[r]) {
            const uint32_t p = d6_perm((uint32_t)r_near++, (uint32_t)n_near, key_sel);
            if (p < (uint32_t)k) choice[p] = in_idx[r];
          } else {
            choice[k + r_far++] = in_idx[r];
          }
        }
      } else {          /* choice(arange(len(points)), N, replace=False) */
        for (int r = 0; r < n_in; ++r) {
          const uint32_t p = d6_perm((uint32_t)r, (uint32_t)n_in, key_sel);
          if (p < (uint32_t)N) choice[p] = in_idx[r];
        }
      }
    } else {            /* arange(len(points)) + extra_choice */
      const int extra = N - n_in;
      for (int r = 0; r < n_in; ++r) choice[r] = in_idx[r];
      if (extra > n_in) {
        for (int e = 0; e < extra; ++e) choice[n_in + e] = in_idx[d6_randint((uint32_t)e, (uint32_t)n_in, key_extra)];
      } else {
        for (int r = 0; r < n_in; ++r) {
          const uint32_t p = d6_perm((uint32_t)r, (uint32_t)n_in, key_sel);
          if (p < (uint32_t)extra) choice[n_in + p] = in_idx[r];
        }
      }
    }
    for (int slot = 0; slot < N; ++slot) {  /* np.random.shuffle(choice); points[choice]; batch-index pad */
      const uint32_t pos = d6_perm((uint32_t)slot, (uint32_t)N, key_shuffle);
      float *dst = out + (size_t)pos * (1 + c);
      const float *src = raw + (size_t)(lo + choice[slot]) * c;
      dst[0] = (float)s;
      for (int j = 0; j < c; ++j) dst[1 + j] = src[j];
    }
    free(in_idx); free(is_near); free(choice);
  }
  return 0;
}

/* the keyed bijection itself, for the unit tests (bijectivity, key sensitivity) */
ORACLE_API void det6d_oracle_perm(uint32_t n, uint64_t seed, uint32_t scene, uint32_t purpose, uint32_t *out) {
  const uint32_t key = d6_stream_key(seed, scene, purpose);
  for (uint32_t x = 0; x < n; ++x) out[x] = d6_perm(x, n, key);
}

/* the CPU restatement needs no scratch */
ORACLE_API int64_t det6d_oracle_prepare_points_workspace_bytes(int b, int total_raw) { (void)b; (void)total_raw; return 0; }

/* ------------------------------------------------------------------------------------------
 * Output consumer: detections -> KITTI annotation fields, float32.
 *   boxes3d_lidar_to_kitti_camera       core/pcdet/utils/box_utils.py:196-212
 *   boxes3d_to_corners3d_kitti_camera   box_utils.py:215-258 (bottom_center=True)
 *   boxes3d_kitti_camera_to_imageboxes  box_utils.py:261-281
 *   Calibration.lidar_to_rect/rect_to_img  core/pcdet/utils/calibration_kitti.py:64-83
 *   alpha = -arctan2(-y, x) + ry        core/pcdet/datasets/kitti/kitti_dataset.py:319
 * Scalar form with every dot product as an ascending fma chain and the shared det6d_math.h
 * sincos / atan2 (the NumPy reference uses BLAS and libm; oracle/annos.py is the literal NumPy
 * restatement that is pinned to the reference, and this function is checked against it to 1e-4).
 * ---------------------------------------------------------------------------------------- */
ORACLE_API int det6d_oracle_kitti_annos(int total, const float *boxes, int ld, const int *scene_of,
                                        const float *calib, float *annos_out) {
  for (int t = 0; t < total; ++t) {
    const float *b = boxes + (size_t)t * ld;
    const float *c = calib + (size_t)scene_of[t] * 28;
    const float *M = c, *P = c + 12;
    const float img_h = c[24], img_w = c[25];
    const float x = b[0], y = b[1], l = b[3], w = b[4], h = b[5], heading = b[6];
    const float z = b[2] - h / 2.f;
    float cam[3];
    for (int j = 0; j < 3; ++j) cam[j] = D6_FMA(z, M[6 + j], D6_FMA(y, M[3 + j], x * M[j])) + M[9 + j];
    const float ry = -heading - 1.57079632679489661923f;
    float sn, cs;
    d6_sincosf(ry, &sn, &cs);
    /* reference corner tables (box_utils.py:230-234) */
    const float sx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, sz[8] = {1, -1, -1, 1, 1, -1, -1, 1};
    float u0 = 3.0e38f, v0 = 3.0e38f, u1 = -3.0e38f, v1 = -3.0e38f;
    for (int k = 0; k < 8; ++k) {
      const float xc = sx[k] * l / 2.f, zc = sz[k] * w / 2.f, yc = k < 4 ? 0.f : -h;
      const float px = cam[0] + D6_FMA(zc, sn, xc * cs);
      const float py = cam[1] + yc;
      const float pz = cam[2] + D6_FMA(zc, cs, -xc * sn);
      const float hu = D6_FMA(pz, P[2], D6_FMA(py, P[1], px * P[0])) + P[3];
      const float hv = D6_FMA(pz, P[6], D6_FMA(py, P[5], px * P[4])) + P[7];
      const float u = hu / pz, v = hv / pz;
      u0 = d6_fminf(u0, u); u1 = d6_fmaxf(u1, u);
      v0 = d6_fminf(v0, v); v1 = d6_fmaxf(v1, v);
    }
    if (img_w > 0.f) {
      u0 = d6_fminf(d6_fmaxf(u0, 0.f), img_w - 1.f); u1 = d6_fminf(d6_fmaxf(u1, 0.f), img_w - 1.f);
      v0 = d6_fminf(d6_fmaxf(v0, 0.f), img_h - 1.f); v1 = d6_fminf(d6_fmaxf(v1, 0.f), img_h - 1.f);
    }
    float *o = annos_out + (size_t)t * 12;
    o[0] = cam[0]; o[1] = cam[1]; o[2] = cam[2];
    o[3] = l; o[4] = h; o[5] = w; o[6] = ry;
    o[7] = u0; o[8] = v0; o[9] = u1; o[10] = v1;
    o[11] = -d6_atan2f(-y, x) + ry;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * SlopeAug geometry, scalar mirrors of det6d_make_slope / det6d_boxes9_corners.
 *   random_global_make_slope (non-smooth branch)  core/pcdet/datasets/augmentor/augmentor_utils.py:670-694
 *   boxes3d_to_corners_3d                          core/pcdet/utils/box_utils.py:57-71
 * oracle/slope.py is the literal NumPy/scipy restatement pinned to the reference's own output;
 * these functions are checked against it to 1e-6 and against the HIP kernels bit for bit.
 * ---------------------------------------------------------------------------------------- */
static double o_sgn(double v) { return (double)((v > 0.0) - (v < 0.0)); }
/* NumPy stores the intermediate results into the float32 array: force the rounding (gcc 11 -O3's SLP
 * vectoriser was seen to drop a (double)(float) round trip here) */
static __attribute__((noinline)) double o_f32(double v) { volatile float f = (float)v; return (double)f; }
static void o_rot3(const double *R, double x, double y, double z, double *o) {
  for (int j = 0; j < 3; ++j) o[j] = fma(z, R[3 * j + 2], fma(y, R[3 * j + 1], x * R[3 * j]));
}
/* limit_period runs in float32 in the reference (check_numpy_to_torch casts to float): each op rounds to float */
static double o_wrap_pi(double v) {
  const float p = 6.283185307179586f, x = (float)v;
  const float q = floorf(x / p + 0.5f);
  const float t = q * p;
  return (double)(x - t);
}

ORACLE_API int det6d_oracle_make_slope(int n_points, float *points, int ld, int n_boxes, double *boxes9,
                                       const double *params) {
  const double *pivot = params, *R = params + 3;
  const double k = params[12], x0 = params[0], y0 = params[1], side = params[13];
  for (int i = 0; i < n_points; ++i) {
    float *p = points + (size_t)i * ld;
    const double x = p[0], y = p[1];
    if (o_sgn(k * (x - x0) + y0 - y) == side) continue;
    const double fx = o_f32(x - pivot[0]), fy = o_f32(y - pivot[1]), fz = o_f32((double)p[2] - pivot[2]);
    double r[3];
    o_rot3(R, fx, fy, fz, r);
    p[0] = (float)o_f32(o_f32(r[0]) + pivot[0]);
    p[1] = (float)o_f32(o_f32(r[1]) + pivot[1]);
    p[2] = (float)o_f32(o_f32(r[2]) + pivot[2]);
  }
  for (int i = 0; i < n_boxes; ++i) {
    double *b = boxes9 + (size_t)i * 9;
    if (o_sgn(k * (b[0] - x0) + y0 - b[1]) != side) {
      double r[3];
      o_rot3(R, b[0] - pivot[0], b[1] - pivot[1], b[2] - pivot[2], r);
      b[0] = r[0] + pivot[0]; b[1] = r[1] + pivot[1]; b[2] = r[2] + pivot[2];
      b[7] += params[14];
      b[8] += params[15];
    }
    b[6] = o_wrap_pi(b[6]); b[7] = o_wrap_pi(b[7]); b[8] = o_wrap_pi(b[8]);
  }
  return 0;
}

ORACLE_API int det6d_oracle_boxes9_corners(int n_boxes, const double *boxes9, double *corners) {
  static const double tx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, ty[8] = {1, -1, -1, 1, 1, -1, -1, 1},
                      tz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
  for (int i = 0; i < n_boxes; ++i) {
    const double *b = boxes9 + (size_t)i * 9;
    const double cz = cos(b[6]), sz = sin(b[6]), cy = cos(b[7]), sy = sin(b[7]), cx = cos(b[8]), sx = sin(b[8]);
    const double R[9] = {cy * cz, -cy * sz, sy,
                         cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy,
                         sx * sz - cx * sy * cz, sx * cz + cx * sy * sz, cx * cy};
    for (int k = 0; k < 8; ++k) {
      double r[3];
      o_rot3(R, tx[k] * b[3] * 0.5, ty[k] * b[4] * 0.5, tz[k] * b[5] * 0.5, r);
      double *o = corners + ((size_t)i * 8 + k) * 3;
      o[0] = r[0] + b[0]; o[1] = r[1] + b[1]; o[2] = r[2] + b[2];
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * KITTI / SlopedKITTI evaluator: scalar mirrors of det6d_eval_overlaps / _match / _reduce.
 *   image_box_overlap        core/pcdet/datasets/kitti/kitti_object_eval_python/eval.py:78-113
 *   rotate_iou_gpu_eval      .../rotate_iou.py (via include/det6d_riou.h)
 *   d3_box_overlap(_kernel)  eval.py:121-155
 *   d9_box_matching_score    core/pcdet/datasets/slopedkitti/kitti_object_eval_python/eval.py:159-193
 *   compute_statistics_jit   eval.py:160-275 ; fused_compute_statistics eval.py:289-342
 * The loops are written the way the reference writes them (explicit assigned / ignored_threshold lists,
 * the three-way if / elif of the matcher, delta list for the orientation similarity).
 * ---------------------------------------------------------------------------------------- */
#include "../include/det6d_riou.h"

/* same layout as det6d_eval_match_args in include/det6d_ops.h (host pointers) */
typedef struct det6d_eval_match_args {
  int n_frames, n_thresh, metric, compute_aos, dt_f32;
  double min_overlap;
  const double *thresholds;
  const int *dt_off, *gt_off, *dc_off;
  const int64_t *pair_off;
  const double *overlaps;
  const double *gt_alpha;
  const double *dt_bbox, *dt_alpha, *dt_score;
  const int *ignored_gt, *ignored_dt;
  const double *dc_bbox;
  unsigned char *workspace;
  double *stats;
  double *tp_scores; int *tp_count; int *gt_of_tp;
} det6d_eval_match_args;

static double o_store(double v, int f32) { return f32 ? o_f32(v) : v; }

static double o_image_overlap(const double *b, const double *q, int dt_f32, int criterion) {
  const double q_area = (q[2] - q[0]) * (q[3] - q[1]);
  const double iw = fmin(b[2], q[2]) - fmax(b[0], q[0]);
  if (iw > 0) {
    const double ih = fmin(b[3], q[3]) - fmax(b[1], q[1]);
    if (ih > 0) {
      double b_area = (b[2] - b[0]) * (b[3] - b[1]);
      if (dt_f32) b_area = o_f32(o_f32((float)b[2] - (float)b[0]) * o_f32((float)b[3] - (float)b[1]));
      double ua = 1.0;
      if (criterion == -1) ua = b_area + q_area - iw * ih;
      else if (criterion == 0) ua = b_area;
      else if (criterion == 1) ua = q_area;
      return iw * ih / ua;
    }
  }
  return 0.0;
}

ORACLE_API int det6d_oracle_eval_overlaps(int metric, int n_frames, const int *dt_off, const int *gt_off,
                                          const int64_t *pair_off, int64_t n_pairs, const double *dt_boxes,
                                          const double *gt_boxes, int dt_f32, double *overlaps) {
  const int ncol = metric == 0 ? 4 : metric == 1 ? 5 : metric == 2 ? 7 : 9;
  (void)n_pairs;
  for (int f = 0; f < n_frames; ++f) {
    const int n_dt = dt_off[f + 1] - dt_off[f], n_gt = gt_off[f + 1] - gt_off[f];
    double *out = overlaps + pair_off[f];
    for (int j = 0; j < n_dt; ++j)
      for (int i = 0; i < n_gt; ++i) {
        const double *b = dt_boxes + (size_t)(dt_off[f] + j) * ncol, *q = gt_boxes + (size_t)(gt_off[f] + i) * ncol;
        double r = 0.0;
        if (metric == 0) {
          r = o_store(o_image_overlap(b, q, dt_f32, -1), dt_f32);
        } else if (metric == 1) {
          const float qb[5] = {(float)q[0], (float)q[1], (float)q[2], (float)q[3], (float)q[4]};
          const float bb[5] = {(float)b[0], (float)b[1], (float)b[2], (float)b[3], (float)b[4]};
          r = (double)d6_riou_eval(qb, bb, -1);
        } else if (metric == 2) {
          const float qb[5] = {(float)q[0], (float)q[2], (float)q[3], (float)q[5], (float)q[6]};
          const float bb[5] = {(float)b[0], (float)b[2], (float)b[3], (float)b[5], (float)b[6]};
          const double rinc = (double)d6_riou_eval(qb, bb, 2);
          if (rinc > 0) {
            const double b_top = dt_f32 ? o_f32((float)b[1] - (float)b[4]) : b[1] - b[4];
            const double iw = fmin(b[1], q[1]) - fmax(b_top, q[1] - q[4]);
            if (iw > 0) {
              const double area1 = dt_f32 ? o_f32(o_f32((float)b[3] * (float)b[4]) * (float)b[5]) : b[3] * b[4] * b[5];
              const double area2 = q[3] * q[4] * q[5];
              const double inc = iw * rinc;
              r = o_store(inc / (area1 + area2 - inc), dt_f32);
            }
          }
        } else {
          const double dx = b[0] - q[0], dy = b[1] - q[1], dz = b[2] - q[2];
          const double dist = sqrt(dx * dx + dy * dy + dz * dz);
          r = o_store(2.0 - 2.0 * (1.0 / (1.0 + exp(-dist))), dt_f32);
        }
        out[(size_t)j * n_gt + i] = r;
      }
  }
  return 0;
}

ORACLE_API int det6d_oracle_eval_match(const det6d_eval_match_args *a) {
  const int T = a->n_thresh > 0 ? a->n_thresh : 1;
  const int compute_fp = a->n_thresh > 0;
  const double NO_DETECTION = -10000000;
  for (int f = 0; f < a->n_frames; ++f) {
    const int d0 = a->dt_off[f], det_size = a->dt_off[f + 1] - d0;
    const int g0 = a->gt_off[f], gt_size = a->gt_off[f + 1] - g0;
    const double *overlaps = a->overlaps + a->pair_off[f];
    char *assigned_detection = (char *)malloc((size_t)det_size + 1);
    char *ignored_threshold = (char *)malloc((size_t)det_size + 1);
    double *delta = (double *)malloc(sizeof(double) * (size_t)(gt_size + 1));
    for (int t = 0; t < T; ++t) {
      const double thresh = compute_fp ? a->thresholds[t] : 0.0;
      for (int j = 0; j < det_size; ++j) {
        assigned_detection[j] = 0;
        ignored_threshold[j] = compute_fp && a->dt_score[d0 + j] < thresh;
      }
      if (!compute_fp && a->gt_of_tp) for (int j = 0; j < det_size; ++j) a->gt_of_tp[d0 + j] = -1;
      int tp = 0, fp = 0, fn = 0, thresh_idx = 0, delta_idx = 0;
      double similarity = 0;
      for (int i = 0; i < gt_size; ++i) {
        if (a->ignored_gt[g0 + i] == -1) continue;
        int det_idx = -1, assigned_ignored_det = 0;
        double valid_detection = NO_DETECTION, max_overlap = 0;
        for (int j = 0; j < det_size; ++j) {
          if (a->ignored_dt[d0 + j] == -1) continue;
          if (assigned_detection[j]) continue;
          if (ignored_threshold[j]) continue;
          const double overlap = overlaps[(size_t)j * gt_size + i], dt_score = a->dt_score[d0 + j];
          if (!compute_fp && overlap > a->min_overlap && dt_score > valid_detection) {
            det_idx = j; valid_detection = dt_score;
          } else if (compute_fp && overlap > a->min_overlap && (overlap > max_overlap || assigned_ignored_det) &&
                     a->ignored_dt[d0 + j] == 0) {
            max_overlap = overlap; det_idx = j; valid_detection = 1; assigned_ignored_det = 0;
          } else if (compute_fp && overlap > a->min_overlap && valid_detection == NO_DETECTION &&
                     a->ignored_dt[d0 + j] == 1) {
            det_idx = j; valid_detection = 1; assigned_ignored_det = 1;
          }
        }
        if (valid_detection == NO_DETECTION && a->ignored_gt[g0 + i] == 0) {
          fn += 1;
        } else if (valid_detection != NO_DETECTION && (a->ignored_gt[g0 + i] == 1 || a->ignored_dt[d0 + det_idx] == 1)) {
          assigned_detection[det_idx] = 1;
        } else if (valid_detection != NO_DETECTION) {
          tp += 1;
          if (!compute_fp) {
            a->tp_scores[g0 + thresh_idx] = a->dt_score[d0 + det_idx];
            if (a->gt_of_tp) a->gt_of_tp[d0 + det_idx] = i;
          }
          thresh_idx += 1;
          if (a->compute_aos) delta[delta_idx++] = a->gt_alpha[g0 + i] - a->dt_alpha[d0 + det_idx];
          assigned_detection[det_idx] = 1;
        }
      }
      if (!compute_fp) { a->tp_count[f] = thresh_idx; continue; }
      for (int j = 0; j < det_size; ++j)
        if (!(assigned_detection[j] || a->ignored_dt[d0 + j] == -1 || a->ignored_dt[d0 + j] == 1 || ignored_threshold[j])) fp += 1;
      int nstuff = 0;
      if (a->metric == 0) {
        for (int c = a->dc_off[f]; c < a->dc_off[f + 1]; ++c)
          for (int j = 0; j < det_size; ++j) {
            if (assigned_detection[j]) continue;
            if (a->ignored_dt[d0 + j] == -1 || a->ignored_dt[d0 + j] == 1) continue;
            if (ignored_threshold[j]) continue;
            const double o = o_store(o_image_overlap(a->dt_bbox + (size_t)(d0 + j) * 4, a->dc_bbox + (size_t)c * 4, a->dt_f32, 0), a->dt_f32);
            if (o > a->min_overlap) { assigned_detection[j] = 1; nstuff += 1; }
          }
      }
      fp -= nstuff;
      if (a->compute_aos) {
        double sum = 0;
        for (int k = 0; k < delta_idx; ++k) sum += (1.0 + cos(delta[k])) / 2.0;
        similarity = (tp > 0 || fp > 0) ? sum : -1;
      }
      double *st = a->stats + ((size_t)f * T + t) * 4;
      st[0] = tp; st[1] = fp; st[2] = fn; st[3] = similarity;
    }
    free(assigned_detection); free(ignored_threshold); free(delta);
  }
  return 0;
}

ORACLE_API int det6d_oracle_eval_reduce(int n_frames, int n_thresh, const double *stats, double *pr) {
  for (int t = 0; t < n_thresh; ++t)
    for (int c = 0; c < 4; ++c) {
      double acc = 0;
      for (int f = 0; f < n_frames; ++f) {
        const double v = stats[((size_t)f * n_thresh + t) * 4 + c];
        if (c == 3 && v == -1) continue;
        acc += v;
      }
      pr[t * 4 + c] = acc;
    }
  return 0;
}
