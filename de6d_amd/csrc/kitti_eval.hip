// kitti_eval.hip — the KITTI / SlopedKITTI detection evaluator on the device.
//
// Reference: core/pcdet/datasets/kitti/kitti_object_eval_python/{eval.py, rotate_iou.py} and the
// slopedkitti copy (eval.py adds metric 3 = d9_box_matching_score and the ATE/ASE/AOE sums).  There,
// only the rotated IoU runs on the GPU (numba.cuda), in "parts" of 50-100 frames whose cross-frame
// pairs are computed and thrown away, and the matching statistics run on the host under numba.jit.
// Here everything per-pair and per-(frame, threshold) runs on the GPU:
//   eval_overlaps_kernel   one thread per (detection, ground truth) pair of the SAME frame, all frames of
//                          the split in one launch (ragged layout, no cross-frame pairs):
//                            metric 0  image_box_overlap        eval.py:78-113
//                            metric 1  bev_box_overlap          eval.py:116-118 + rotate_iou.py
//                            metric 2  d3_box_overlap(_kernel)  eval.py:121-155
//                            metric 3  d9_box_matching_score    slopedkitti eval.py:159-193 (score_type 0)
//   eval_match_kernel      compute_statistics_jit (eval.py:160-275): one thread per frame (pass A,
//                          compute_fp = False: the scores of the matched detections) or per
//                          (frame, score threshold) (pass B = fused_compute_statistics, eval.py:289-342)
// NumPy / numba dtype behaviour is part of the result (detections written by generate_prediction_dicts
// are float32, labels float64); `dt_f32` reproduces it.
#include "common.h"
#include "../../include/det6d_riou.h"

namespace {

__device__ __forceinline__ int find_frame(const int64_t *pair_off, int n_frames, int64_t t) {
  int lo = 0, hi = n_frames;  // largest f with pair_off[f] <= t
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pair_off[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ double stored(double v, int as_f32) { return as_f32 ? (double)(float)v : v; }

__device__ double image_overlap(const double *b, const double *q, int dt_f32, int criterion) {
  const double q_area = (q[2] - q[0]) * (q[3] - q[1]);
  const double iw = fmin(b[2], q[2]) - fmax(b[0], q[0]);
  if (!(iw > 0)) return 0.0;
  const double ih = fmin(b[3], q[3]) - fmax(b[1], q[1]);
  if (!(ih > 0)) return 0.0;
  const double b_area = dt_f32 ? (double)((float)((float)b[2] - (float)b[0]) * (float)((float)b[3] - (float)b[1]))
                               : (b[2] - b[0]) * (b[3] - b[1]);
  double ua;
  if (criterion == -1) ua = b_area + q_area - iw * ih;
  else if (criterion == 0) ua = b_area;
  else if (criterion == 1) ua = q_area;
  else ua = 1.0;
  return iw * ih / ua;
}

__global__ __launch_bounds__(256) void eval_overlaps_kernel(int metric, int n_frames, const int *dt_off, const int *gt_off,
                                                           const int64_t *pair_off, const double *dt_boxes,
                                                           const double *gt_boxes, int dt_f32, double *out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= pair_off[n_frames]) return;
  const int f = find_frame(pair_off, n_frames, t);
  const int n_gt = gt_off[f + 1] - gt_off[f];
  const int64_t local = t - pair_off[f];
  const int j = (int)(local / n_gt), i = (int)(local % n_gt);   // overlaps[j = detection, i = ground truth]
  const int ncol = metric == 0 ? 4 : metric == 1 ? 5 : metric == 2 ? 7 : 9;
  const double *b = dt_boxes + (size_t)(dt_off[f] + j) * ncol;
  const double *q = gt_boxes + (size_t)(gt_off[f] + i) * ncol;
  double r;
  if (metric == 0) {
    r = stored(image_overlap(b, q, dt_f32, -1), dt_f32);
  } else if (metric == 1) {
    const float qb[5] = {(float)q[0], (float)q[1], (float)q[2], (float)q[3], (float)q[4]};
    const float bb[5] = {(float)b[0], (float)b[1], (float)b[2], (float)b[3], (float)b[4]};
    r = (double)d6_riou_eval(qb, bb, -1);
  } else if (metric == 2) {
    // camera boxes [x, y, z, l, h, w, ry]: BEV rectangle (x, z, l, w, ry), height along -y
    const float qb[5] = {(float)q[0], (float)q[2], (float)q[3], (float)q[5], (float)q[6]};
    const float bb[5] = {(float)b[0], (float)b[2], (float)b[3], (float)b[5], (float)b[6]};
    const double inter_bev = (double)d6_riou_eval(qb, bb, 2);
    r = 0.0;
    if (inter_bev > 0) {
      const double b_top = dt_f32 ? (double)((float)b[1] - (float)b[4]) : b[1] - b[4];
      const double ih = fmin(b[1], q[1]) - fmax(b_top, q[1] - q[4]);
      if (ih > 0) {
        const double v1 = dt_f32 ? (double)((float)b[3] * (float)b[4] * (float)b[5]) : b[3] * b[4] * b[5];
        const double v2 = q[3] * q[4] * q[5];
        const double inc = ih * inter_bev;
        r = stored(inc / (v1 + v2 - inc), dt_f32);
      }
    }
  } else {
    const double dx = b[0] - q[0], dy = b[1] - q[1], dz = b[2] - q[2];
    const double dist = sqrt(dx * dx + dy * dy + dz * dz);
    r = stored(2.0 - 2.0 * (1.0 / (1.0 + exp(-dist))), dt_f32);
  }
  out[t] = r;
}

constexpr double kNoDetection = -10000000.0;

// one thread = one (frame, threshold) of pass B, or one frame of pass A
__global__ __launch_bounds__(64) void eval_match_kernel(const det6d_eval_match_args a) {
  const int T = a.n_thresh > 0 ? a.n_thresh : 1;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= a.n_frames * T) return;
  const int f = tid / T, tt = tid % T;
  const bool compute_fp = a.n_thresh > 0;
  const double thresh = compute_fp ? a.thresholds[tt] : 0.0;
  const int d0 = a.dt_off[f], n_dt = a.dt_off[f + 1] - d0;
  const int g0 = a.gt_off[f], n_gt = a.gt_off[f + 1] - g0;
  const double *ov = a.overlaps + a.pair_off[f];
  const int *ign_gt = a.ignored_gt + g0, *ign_dt = a.ignored_dt + d0;
  const double *score = a.dt_score + d0;
  unsigned char *assigned = a.workspace + (size_t)d0 * T + (size_t)tt * n_dt;
  for (int j = 0; j < n_dt; ++j) assigned[j] = 0;
  if (!compute_fp && a.gt_of_tp)
    for (int j = 0; j < n_dt; ++j) a.gt_of_tp[d0 + j] = -1;

  int tp = 0, fp = 0, fn = 0, n_scores = 0;
  double similarity = 0.0;
  for (int i = 0; i < n_gt; ++i) {
    if (ign_gt[i] == -1) continue;
    int det_idx = -1;
    double valid_detection = kNoDetection, max_overlap = 0.0;
    bool assigned_ignored_det = false;
    for (int j = 0; j < n_dt; ++j) {
      if (ign_dt[j] == -1 || assigned[j] || (compute_fp && score[j] < thresh)) continue;
      const double overlap = ov[(size_t)j * n_gt + i];
      if (!(overlap > a.min_overlap)) continue;
      if (!compute_fp) {
        if (score[j] > valid_detection) { det_idx = j; valid_detection = score[j]; }
      } else if ((overlap > max_overlap || assigned_ignored_det) && ign_dt[j] == 0) {
        max_overlap = overlap; det_idx = j; valid_detection = 1.0; assigned_ignored_det = false;
      } else if (valid_detection == kNoDetection && ign_dt[j] == 1) {
        det_idx = j; valid_detection = 1.0; assigned_ignored_det = true;
      }
    }
    if (valid_detection == kNoDetection) {
      if (ign_gt[i] == 0) ++fn;
    } else if (ign_gt[i] == 1 || ign_dt[det_idx] == 1) {
      assigned[det_idx] = 1;
    } else {
      ++tp;
      if (!compute_fp) {
        a.tp_scores[g0 + n_scores] = score[det_idx];
        if (a.gt_of_tp) a.gt_of_tp[d0 + det_idx] = i;
      }
      ++n_scores;
      if (a.compute_aos) similarity += (1.0 + cos(a.gt_alpha[g0 + i] - a.dt_alpha[d0 + det_idx])) / 2.0;
      assigned[det_idx] = 1;
    }
  }
  if (!compute_fp) {
    a.tp_count[f] = n_scores;
    return;
  }
  for (int j = 0; j < n_dt; ++j)
    if (!(assigned[j] || ign_dt[j] == -1 || ign_dt[j] == 1 || score[j] < thresh)) ++fp;
  if (a.metric == 0) {  // detections sitting on DontCare regions are not false positives
    int nstuff = 0;
    for (int c = a.dc_off[f]; c < a.dc_off[f + 1]; ++c)
      for (int j = 0; j < n_dt; ++j) {
        if (assigned[j] || ign_dt[j] == -1 || ign_dt[j] == 1 || score[j] < thresh) continue;
        const double o = stored(image_overlap(a.dt_bbox + (size_t)(d0 + j) * 4, a.dc_bbox + (size_t)c * 4, a.dt_f32, 0), a.dt_f32);
        if (o > a.min_overlap) { assigned[j] = 1; ++nstuff; }
      }
    fp -= nstuff;
  }
  if (a.compute_aos && !(tp > 0 || fp > 0)) similarity = -1.0;
  double *st = a.stats + ((size_t)f * T + tt) * 4;
  st[0] = tp; st[1] = fp; st[2] = fn; st[3] = a.compute_aos ? similarity : 0.0;
}

__global__ void eval_reduce_kernel(int n_frames, int n_thresh, const double *stats, double *pr) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n_thresh * 4) return;
  const int t = tid / 4, c = tid % 4;
  double acc = 0.0;
  for (int f = 0; f < n_frames; ++f) {
    const double v = stats[((size_t)f * n_thresh + t) * 4 + c];
    if (c == 3 && v == -1.0) continue;
    acc += v;
  }
  pr[tid] = acc;
}

}  // namespace

DET6D_API int det6d_eval_match(const det6d_eval_match_args *a, det6d_stream_t stream) {
  if (!a || a->n_frames < 0 || a->n_thresh < 0 || a->metric < 0 || a->metric > 3) return DET6D_EINVAL;
  if (a->n_frames == 0) return DET6D_OK;
  if (!a->dt_off || !a->gt_off || !a->dc_off || !a->pair_off || !a->workspace || !a->ignored_gt || !a->ignored_dt ||
      !a->dt_score)
    return DET6D_EINVAL;
  if (a->n_thresh > 0 ? (!a->thresholds || !a->stats || !a->dt_bbox) : (!a->tp_scores || !a->tp_count)) return DET6D_EINVAL;
  if (a->compute_aos && (!a->gt_alpha || !a->dt_alpha)) return DET6D_EINVAL;
  const int total = a->n_frames * (a->n_thresh > 0 ? a->n_thresh : 1);
  hipLaunchKernelGGL(eval_match_kernel, dim3(det6d_divup(total, 64)), dim3(64), 0, (hipStream_t)stream, *a);
  return det6d_check_launch("det6d_eval_match");
}

DET6D_API int det6d_eval_reduce(int n_frames, int n_thresh, const double *stats, double *pr, det6d_stream_t stream) {
  if (n_frames < 0 || n_thresh < 0) return DET6D_EINVAL;
  if (n_thresh == 0) return DET6D_OK;
  if (!pr || (n_frames > 0 && !stats)) return DET6D_EINVAL;
  hipLaunchKernelGGL(eval_reduce_kernel, dim3(det6d_divup(n_thresh * 4, 64)), dim3(64), 0, (hipStream_t)stream, n_frames,
                     n_thresh, stats, pr);
  return det6d_check_launch("det6d_eval_reduce");
}

DET6D_API int det6d_eval_overlaps(int metric, int n_frames, const int *dt_off, const int *gt_off, const int64_t *pair_off,
                                  int64_t n_pairs, const double *dt_boxes, const double *gt_boxes, int dt_f32,
                                  double *overlaps, det6d_stream_t stream) {
  if (metric < 0 || metric > 3 || n_frames < 0 || n_pairs < 0) return DET6D_EINVAL;
  if (n_frames == 0 || n_pairs == 0) return DET6D_OK;
  if (!dt_off || !gt_off || !pair_off || !dt_boxes || !gt_boxes || !overlaps) return DET6D_EINVAL;
  hipLaunchKernelGGL(eval_overlaps_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     metric, n_frames, dt_off, gt_off, pair_off, dt_boxes, gt_boxes, dt_f32, overlaps);
  return det6d_check_launch("det6d_eval_overlaps");
}
