/*
 * det6d_ops.h — C ABI of libdet6d_hip.so, the MI355X (gfx950) drop-in for the two compiled
 * extension modules on the Det6D inference path of HITSZ-NRSL/De6D:
 *
 *   pointnet2_batch_cuda  (core/pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:11-30)
 *   iou3d_nms_cuda        (core/pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17)
 *
 * Conventions (they replace the reference's at::Tensor / exit(-1) conventions,
 * ball_query.cpp:17-29, sampling_gpu.cu:261-265):
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns all buffers (outputs and scratch), exactly like the reference
 *     (pointnet2_utils.py:25-26,294,323-324; iou3d_nms_utils.py:97);
 *   - all tensors are dense, row-major, fp32 / int32 unless stated;
 *   - `stream` is a hipStream_t (NULL = the null stream); calls are asynchronous and
 *     graph-capturable unless documented otherwise;
 *   - return value: DET6D_OK (0) or a negative DET6D_E* code; nothing ever calls exit().
 *
 * The same signatures with the prefix `det6d_oracle_` and HOST pointers (no stream argument)
 * are exported by oracle/libdet6d_oracle.so, the CPU restatement used only by tests.
 */
#ifndef DET6D_OPS_H
#define DET6D_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t *det6d_stream_t; /* == hipStream_t */

#define DET6D_OK 0
#define DET6D_EINVAL (-1)   /* bad size / null pointer / unsupported shape */
#define DET6D_ELAUNCH (-2)  /* hipGetLastError() != hipSuccess after the launch */
#define DET6D_ENOGPU (-3)   /* no HIP device visible */

/* library identification: returns a static string "det6d-hip gfx950 <abi-version>" */
const char *det6d_version(void);
/* last HIP error string seen by this library on the calling thread ("" if none) */
const char *det6d_last_error(void);

/* ------------------------------------------------------------------ sampling ------------- */

/* D-FPS. Replaces farthest_point_sampling_wrapper(b,n,m,xyz,temp,idx)
 * (sampling.cpp:40-50 -> sampling_gpu.cu:101-266).
 *   xyz (B,N,3) f32; temp (B,N) f32 scratch, caller pre-fills 1e10, clobbered; idx (B,M) i32.
 * idx[:,0] = 0; ties resolved exactly like the reference's strided scan + smem tree. */
int det6d_fps(int b, int n, int m, const float *xyz, float *temp, int *idx, det6d_stream_t stream);

/* S-FPS. Replaces furthest_point_sampling_weights_wrapper(b,n,m,xyz,weights,temp,idx)
 * (sampling.cpp:66-79 -> sampling_gpu.cu:419-585).  weights (B,N) f32. */
int det6d_fps_weights(int b, int n, int m, const float *xyz, const float *weights, float *temp,
                      int *idx, det6d_stream_t stream);

/* Replaces gather_points_wrapper(b,c,n,npoints,points,idx,out) (sampling_gpu.cu:16-52).
 *   points (B,C,N); idx (B,M) i32; out (B,C,M). */
int det6d_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                        float *out, det6d_stream_t stream);

/* Replaces gather_points_grad_wrapper (sampling_gpu.cu:54-90): grad_points (B,C,N) += scatter. */
int det6d_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                             const int *idx, float *grad_points, det6d_stream_t stream);

/* ------------------------------------------------------------------ ball query ----------- */

/* Replaces ball_query_wrapper (ball_query_gpu.cu:15-51,157-176).
 *   new_xyz (B,M,3); xyz (B,N,3); idx (B,M,nsample) i32, caller zero-fills.
 * The three single-shell entries take nsample <= 4096 (one LDS hit list per wave; the reference has no cap, nobody
 * configures more than 128); det6d_ball_query_pair and the grid form are narrower: see there. */
int det6d_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                     const float *xyz, int *idx, det6d_stream_t stream);

/* Replaces ball_query_cnt_wrapper (ball_query_gpu.cu:93-153). idx_cnt (B,M) i32. */
int det6d_ball_query_cnt(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                         const float *xyz, int *idx_cnt, int *idx, det6d_stream_t stream);

/* Replaces ball_query_dilated_wrapper (ball_query_gpu.cu:53-91,178-198). */
int det6d_ball_query_dilated(int b, int n, int m, float radius_in, float radius_out, int nsample,
                             const float *new_xyz, const float *xyz, int *idx_cnt, int *idx,
                             det6d_stream_t stream);

/* Fused form of two cnt/dilated queries over the same (new_xyz, xyz): shell A accepts
 * rin_a^2 <= d2 < rout_a^2 (rin = 0: plain ball), shell B likewise; results are identical to two
 * separate det6d_ball_query_dilated / _cnt calls on zero-filled idx buffers (empty balls are
 * written as zeros here, so no memset is needed).  ns_a, ns_b <= 128 (DET6D_EINVAL beyond: two single-shell calls).
 * One sweep over the points feeds both groups of an
 * SA layer (pointnet2_modules.py:462-463 loops the groupers over the same inputs). */
int det6d_ball_query_pair(int b, int n, int m, float rin_a, float rout_a, int ns_a, float rin_b,
                          float rout_b, int ns_b, const float *new_xyz, const float *xyz, int *cnt_a,
                          int *idx_a, int *cnt_b, int *idx_b, det6d_stream_t stream);

/* Grid-hashed form of det6d_ball_query_pair for large N (identical results): a uniform (x, y) grid
 * with cell edge >= max(rout_a, rout_b) is built per scene, a centre tests only the points of its
 * 3x3 cell neighbourhood and keeps, per shell, the nsample smallest hit indices (a centre with few candidates walks them
 * in one lane with a sorted per-lane list, a centre with many takes a wave: a short LDS list with a pruning threshold,
 * ranked once at the end: the reference's ascending-index order either way).  workspace:
 * det6d_ball_query_grid_workspace_bytes(b, n) bytes, 16-byte aligned, caller owned.  ns_a, ns_b <= 64
 * (DET6D_EINVAL beyond: use det6d_ball_query_pair).  idx_a (idx_b) must be 16-byte aligned when ns_a (ns_b) is a multiple of 4
 * (index rows leave as 16-byte stores; DET6D_EINVAL otherwise, before anything is queued — an offset view of a larger buffer
 * goes through det6d_ball_query_pair).  det6d_ball_query_grid_supported(n, ns_a, ns_b): 1 when this entry takes
 * the shape (the host asks before it routes a query here instead of through det6d_ball_query_pair / _cnt / _dilated, which
 * have no nsample limit, like the reference: ball_query_gpu.cu:53-130). */
int det6d_ball_query_grid_supported(int n, int ns_a, int ns_b);
int64_t det6d_ball_query_grid_workspace_bytes(int b, int n);
int det6d_ball_query_pair_grid(int b, int n, int m, float rin_a, float rout_a, int ns_a, float rin_b,
                               float rout_b, int ns_b, const float *new_xyz, const float *xyz,
                               void *workspace, int *cnt_a, int *idx_a, int *cnt_b, int *idx_b,
                               det6d_stream_t stream);

/* ------------------------------------------------------------------ grouping ------------- */

/* Replaces group_points_wrapper (group_points_gpu.cu:53-92).
 *   points (B,C,N); idx (B,npoints,nsample) i32; out (B,C,npoints,nsample). */
int det6d_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                       const int *idx, float *out, det6d_stream_t stream);

/* Replaces group_points_grad_wrapper (group_points_gpu.cu:14-51). */
int det6d_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                            const int *idx, float *grad_points, det6d_stream_t stream);

/* ------------------------------------------------------------------ interpolation -------- */

/* Replaces three_nn_wrapper(b,n,m,unknown,known,dist2,idx) (interpolate_gpu.cu:16-82).
 *   unknown (B,n,3); known (B,m,3); dist2 (B,n,3) f32; idx (B,n,3) i32. */
int det6d_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                   int *idx, det6d_stream_t stream);

/* Replaces three_interpolate_wrapper(b,c,m,n,points,idx,weight,out) (interpolate_gpu.cu:84-124).
 *   points (B,C,m); idx (B,n,3); weight (B,n,3); out (B,C,n). */
int det6d_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                            const float *weight, float *out, det6d_stream_t stream);

/* Replaces three_interpolate_grad_wrapper (interpolate_gpu.cu:127-170). */
int det6d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                 const float *weight, float *grad_points, det6d_stream_t stream);

/* ------------------------------------------------------------------ rotated IoU / NMS ---- */

/* Replaces boxes_overlap_bev_gpu / boxes_iou_bev_gpu (iou3d_nms.cpp:49-88,
 * iou3d_nms_kernel.cu:236-265).  boxes (K,7) [x,y,z,dx,dy,dz,heading]; out (num_a,num_b). */
int det6d_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                            float *ans_overlap, det6d_stream_t stream);
int det6d_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                        float *ans_iou, det6d_stream_t stream);

/* Replaces boxes_iou_bev_cpu (iou3d_nms_api.cpp:16, iou3d_cpu.cpp:232-252; Python caller boxes_bev_iou_cpu,
 * iou3d_nms_utils.py:12-29): the reference runs this entry on the CPU over HOST tensors, and so does this one
 * (all three pointers are host memory, no stream; same geometry code as the device kernels). */
int det6d_boxes_iou_bev_cpu(int num_a, const float *boxes_a_host, int num_b, const float *boxes_b_host,
                            float *ans_iou_host);

/* number of uint64 words the mask scratch of det6d_nms* needs: K * ceil(K/64) */
int64_t det6d_nms_mask_words(int boxes_num);

/* Rotated-BEV NMS, fully on device. Replaces nms_gpu (iou3d_nms.cpp:90-136 +
 * iou3d_nms_kernel.cu:267-311): suppression bit-matrix with one wave64 ballot per (box, 64-column block)
 * = one mask word (K x ceil(K/64) waves over the chip), then the greedy scan by a single wave over rows staged in
 * LDS, no host round trip.
 *   boxes (K,7) sorted by descending score; mask (det6d_nms_mask_words(K)) u64 scratch (words of the column blocks
 *   below a row's own block are never read by the scan and are left unwritten);
 *   keep (K) i64 DEVICE, first *num_keep entries valid; num_keep (1) i32 DEVICE. */
int det6d_nms(int boxes_num, const float *boxes, float thresh, uint64_t *mask, int64_t *keep,
              int *num_keep, det6d_stream_t stream);

/* Axis-aligned variant. Replaces nms_normal_gpu (iou3d_nms.cpp:139-186, kernel :313-372). */
int det6d_nms_normal(int boxes_num, const float *boxes, float thresh, uint64_t *mask,
                     int64_t *keep, int *num_keep, det6d_stream_t stream);

/* Reference-shaped convenience for the pybind-style `nms_gpu(boxes, keep_cpu, thr) -> num`:
 * runs det6d_nms on `stream`, synchronises it and copies keep to HOST memory.
 * NOT graph-capturable (it blocks, like the reference does). */
int det6d_nms_to_host(int boxes_num, const float *boxes, float thresh, int64_t *keep_host,
                      int normal, det6d_stream_t stream);

/* ------------------------------------------------------------------ input producer ------- */
/* Raw frames -> the model's `points (B*num_points, 1+C)` tensor [b, x, y, z, feat..] in one launch
 * (SURVEY.md §8 f1).  Replaces, for the eval/test pipeline,
 *   DataProcessor.mask_points_and_boxes_outside_range  datasets/processor/data_processor.py:78-90
 *     (x/y range only: utils/common_utils.py:61-64),
 *   DataProcessor.sample_points                        data_processor.py:145-178,
 *   DatasetTemplate.collate_batch ('points' branch)    datasets/dataset.py:171-176,
 * and moves load_data_to_gpu (models/__init__.py:23-34) in front of them: the caller uploads the RAW
 * frames `raw (total_raw, c)` with `raw_offsets (b+1)` (device, int32).
 * Selection rule = the reference's (all points farther than `near_depth` = 40 m kept while they fit,
 * the rest drawn without replacement from the near ones; short frames padded with duplicates; final
 * shuffle).  The random draws come from keyed bijections (include/det6d_rng.h) instead of numpy's
 * global Mersenne Twister, reproducible from (seed, scene id): scene_ids (b, device int32, may be NULL =
 * position in the batch) names each frame, so a frame is sampled the same way however it is batched
 * or sharded over GPUs.  A scene with no point in range
 * yields zero rows (the reference raises).  n_in_range (b) receives the in-range counts.
 * workspace: det6d_prepare_points_workspace_bytes(b, total_raw) bytes. */
int64_t det6d_prepare_points_workspace_bytes(int b, int total_raw);
int det6d_prepare_points(int b, const int *raw_offsets, const int *scene_ids, int total_raw, int c,
                         const float *raw,
                         float x_min, float y_min, float x_max, float y_max, int num_points,
                         float near_depth, uint64_t seed, void *workspace, float *points_out,
                         int *n_in_range, det6d_stream_t stream);

/* ------------------------------------------------------------------ output consumer ------ */
/* Detections of a batch (LiDAR frame) -> KITTI annotation fields in one launch:
 * the device half of KittiDataset.generate_prediction_dicts (datasets/kitti/kitti_dataset.py:277-351;
 * slopedkitti/kitti_dataset.py:299-379), i.e. box_utils.boxes3d_lidar_to_kitti_camera (box_utils.py:196-212),
 * boxes3d_kitti_camera_to_imageboxes (:261-281) with Calibration.lidar_to_rect / rect_to_img
 * (calibration_kitti.py:64-83), and alpha (kitti_dataset.py:319).
 * boxes (total, ld >= 7) [x,y,z,dx,dy,dz,heading,..]; scene_of (total) int32 = row of `calib`;
 * calib (B, 28) float32 per scene: M = V2C^T R0^T (4x3 row-major), P2 (3x4 row-major), image height,
 * image width (<= 0: no clipping), 2 pad.  annos_out (total, 12): camera box [x,y,z,l,h,w,ry],
 * image box [x1,y1,x2,y2], alpha.  float32; tolerance vs the NumPy reference 1e-4 (BLAS dot order). */
int det6d_kitti_annos(int total, const float *boxes, int ld, const int *scene_of, const float *calib,
                      float *annos_out, det6d_stream_t stream);

/* ------------------------------------------------------------------ SlopeAug geometry ---- */
/* Per-point / per-box half of random_global_make_slope (datasets/augmentor/augmentor_utils.py:670-694,
 * SURVEY.md §8 f4): points (n_points, ld) float32 and boxes9 (n_boxes, 9) float64
 * [x,y,z,dx,dy,dz,rz,ry,rx] are updated IN PLACE.  params (16 doubles, HOST memory, read at call time):
 * pivot xyz [0:3], rotation matrix R row-major [3:12] (= Rotation.from_rotvec(angle).as_matrix()),
 * k = angle_y / (angle_x + 1e-6) [12], sensor side sign(k (0 - x0) + y0) [13], pitch / roll increments
 * (euler 'XYZ' [1], [0]) [14], [15].  Everything with sign(k (x - x0) + y0 - y) != side is rotated about
 * the pivot; all box angles are wrapped to [-pi, pi) (common_utils.limit_period).  NumPy's dtype
 * behaviour is kept (float32 points updated through float64 temporaries).  Tolerance vs NumPy 1e-6. */
int det6d_make_slope(int n_points, float *points, int ld, int n_boxes, double *boxes9, const double *params,
                     det6d_stream_t stream);

/* corners (n_boxes, 8, 3) float64 of 9-D boxes, R = Rx(rx) Ry(ry) Rz(rz) (scipy 'zyx', extrinsic): box_utils.boxes3d_to_corners_3d
 * (core/pcdet/utils/box_utils.py:57-71). */
int det6d_boxes9_corners(int n_boxes, const double *boxes9, double *corners, det6d_stream_t stream);

/* ------------------------------------------------------------------ KITTI evaluator ------ */
/* SURVEY.md §8 f3.  Reference: datasets/kitti/kitti_object_eval_python/{eval.py, rotate_iou.py} and the
 * slopedkitti copy.  Frames are laid out ragged: dt_off / gt_off / dc_off (n_frames + 1, int32) index the
 * concatenated detections / ground truths / DontCare boxes, pair_off (n_frames + 1, int64) the per-frame
 * (n_dt x n_gt) row-major overlap blocks.  All arrays are device memory; boxes are float64 (float32
 * detections converted exactly), `dt_f32` != 0 reproduces the float32 typing numba / NumPy apply to them.
 *
 * det6d_eval_overlaps: overlaps[pair_off[f] + j * n_gt + i] for detection j, ground truth i of frame f.
 *   metric 0  image_box_overlap (eval.py:78-113), boxes (.,4) [x1,y1,x2,y2]
 *   metric 1  bev_box_overlap = rotate_iou_gpu_eval (eval.py:116-118, rotate_iou.py), boxes (.,5) [x,z,l,w,ry]
 *   metric 2  d3_box_overlap (eval.py:121-155), boxes (.,7) camera [x,y,z,l,h,w,ry]
 *   metric 3  d9_box_matching_score, score_type 0 (slopedkitti eval.py:159-193), boxes (.,9) [loc,dims,yaw,pitch,roll] */
int det6d_eval_overlaps(int metric, int n_frames, const int *dt_off, const int *gt_off, const int64_t *pair_off,
                        int64_t n_pairs, const double *dt_boxes, const double *gt_boxes, int dt_f32,
                        double *overlaps, det6d_stream_t stream);

/* compute_statistics_jit (eval.py:160-275) for every frame (x every score threshold): the greedy
 * ground-truth -> detection matching.  n_thresh == 0: pass A (compute_fp = False, thresh = 0): writes the
 * scores of the true positives of frame f to tp_scores[gt_off[f] ..], their number to tp_count[f] and, if
 * gt_of_tp != NULL, the matched ground-truth index per detection (-1 otherwise; slopedkitti eval.py:216,277).
 * n_thresh > 0: pass B (fused_compute_statistics, eval.py:289-342): stats (n_frames, n_thresh, 4) =
 * [tp, fp, fn, similarity] per frame and threshold; det6d_eval_reduce sums them over frames in frame order
 * (similarity only where it is not -1) into pr (n_thresh, 4).
 * workspace: max(1, total_dt * max(1, n_thresh)) bytes. */
typedef struct det6d_eval_match_args {
  int n_frames, n_thresh, metric, compute_aos, dt_f32;
  double min_overlap;
  const double *thresholds;                 /* (n_thresh) */
  const int *dt_off, *gt_off, *dc_off;      /* (n_frames + 1) */
  const int64_t *pair_off;                  /* (n_frames + 1) */
  const double *overlaps;
  const double *gt_alpha;                   /* (total_gt) */
  const double *dt_bbox, *dt_alpha, *dt_score; /* (total_dt, 4), (total_dt), (total_dt) */
  const int *ignored_gt, *ignored_dt;       /* 0 evaluate, 1 ignore, -1 other class (clean_data, eval.py:29-75) */
  const double *dc_bbox;                    /* (total_dc, 4) */
  unsigned char *workspace;
  double *stats;                            /* pass B */
  double *tp_scores; int *tp_count; int *gt_of_tp; /* pass A */
} det6d_eval_match_args;
int det6d_eval_match(const det6d_eval_match_args *args, det6d_stream_t stream);
int det6d_eval_reduce(int n_frames, int n_thresh, const double *stats, double *pr, det6d_stream_t stream);

/* ------------------------------------------------------------------ fused engine ops ----- */
/* These have no 1:1 reference symbol; they implement the Python-level hot loop of
 * _PointnetSAModuleFSBase.forward (pointnet2_modules.py:462-494) and
 * PointHeadBox6DVote.forward (point_head_box6d_vote.py:794-903) as fused kernels. */

/* Pack the reference's flat `points (B*N, 1+3+C)` rows [b,x,y,z,feat..] into the engine's
 * point-major row layout `rows (B,N,ld)` = [x,y,z,feat..,0 pad], ld % 4 == 0.
 * (PointNet2FSMSG.break_up_pc + view/permute, pointnet2_backbone.py:193-224.) */
int det6d_pack_points(int total, int cin, const float *points, int ld, float *rows, float *xyz_out,
                      det6d_stream_t stream);   /* xyz_out (total,3) optional packed copy of the coordinates */

/* FPS as the SA layer uses it (pointnet2_modules.py:376-450), one launch per sampler and nothing
 * else: samples range [lo, hi) of xyz (B, n_total, 3); `scores` == NULL -> d-fps, otherwise s-fps
 * with weights sigmoid(scores[b, k])**gamma computed in the kernel (scores (B, n_total));
 * min-distances start at 1e10 implicitly; the picks + lo are written to idx[b*idx_stride +
 * idx_offset + j].  temp: scratch of temp_bytes bytes, at least (B, hi-lo) x 4 (min-distances of the
 * memory-resident kernel, or the Morton permutation of the pruned samplers used for d-fps on 8192 /
 * 16384 points); with det6d_fps_fused_workspace_bytes(B, hi-lo) bytes, d-fps of 32768 / 65536-point scenes
 * runs register-resident on 2 / 4 cooperating workgroups per scene (csrc/fps_coop.hip) instead of the
 * memory-resident kernel (same picks, ~100x faster). */
long long det6d_fps_fused_workspace_bytes(int b, int n);
/* The cooperative sampler's failure path.  Its 2 / 4 workgroups per scene exchange candidates through memory; a workgroup
 * that waits ~2 s for a partner that was never scheduled gives up, fills the rest of the scene's picks with the (valid)
 * first index and sets an int32 error word inside the workspace.  The word is STICKY: `temp` must be zero-filled once when
 * it is allocated, no launch clears the word, det6d_fps_fused_status clears it after reading it set.
 * det6d_fps_fused_status: after launches of b scenes of n = hi-lo points on `temp`: synchronises `stream`; DET6D_OK, or
 *   DET6D_ELAUNCH when a launch since the last read gave up (its picks are invalid).  Blocks: not capturable.
 * det6d_fps_fused_status_offset: byte offset of that word from `temp` (so that a pipeline can copy it to pinned memory
 *   next to its other results without a synchronisation of its own), -1 when the sampler that (b, n, temp_bytes) selects
 *   cannot fail after its launch.
 * Co-residency: ONE call never asks for more than one workgroup per CU at a time (a call over more scenes than CUs / parts
 * samples them chunk by chunk on its stream), so a call cannot starve itself whatever b is.  Calls in flight at the same
 * time on DIFFERENT streams are the caller's to bound: together they must not ask for more than one workgroup per CU
 * (2 / 4 x scenes each), otherwise partially dispatched launches can starve each other until the time-out
 * (de6d_amd/runtime.py: ScenePipeline keeps them on few enough streams). */
int det6d_fps_fused_status(int b, int n, const float *temp, long long temp_bytes, det6d_stream_t stream);
long long det6d_fps_fused_status_offset(int b, int n, const float *temp, long long temp_bytes);
int det6d_fps_fused(int b, int n_total, int lo, int hi, int m, const float *xyz, const float *scores,
                    float gamma, float *temp, long long temp_bytes, int *idx, int idx_stride, int idx_offset,
                    int idx_bias, det6d_stream_t stream);   /* idx_bias: added to every written index on top of lo */

/* Short stacks of plain pointwise layers over few rows in ONE launch (csrc/mlp_rows.hip): the aggregation + confidence
 * chain of an SA layer (pointnet2_modules.py:580-607), the vote FC and the cls / reg towers of the head
 * (point_head_box6d_vote.py:33-45,157-169).  Input: columns [xcol0, xcol0 + k) of x (rows, ldx), k = layers[0].k.
 * Layer l: y = act(in[0..k) x W[wrow0 .. wrow0 + k)[0..n) + shift); k % 32 == 0; hidden widths n % 32 == 0 and equal to the
 * next layer's k; `out` (rows, ldo) optional for hidden layers, required for the last; 1 or 2 chains over the same input
 * (nlayers[c] layers each, stored back to back in `layers`), at most 4 layers per chain.  Bit for bit the corresponding
 * sequence of det6d_linear calls. */
typedef struct det6d_rows_layer {
  const float *w; int ldw; int wrow0;
  const float *shift;
  int k, n, act;
  float *out; int ldo; int ocol0;
} det6d_rows_layer;
int det6d_mlp_rows(int rows, const float *x, int ldx, int xcol0, int nchains, const int *nlayers,
                   const det6d_rows_layer *layers, det6d_stream_t stream);
/* 1 when det6d_mlp_rows accepts this stack (same checks, incl. the 160 KB of LDS a 32-row tile's two activation buffers
 * may take), else 0 — asked by the host before it routes a stack here instead of through one det6d_linear per layer. */
int det6d_mlp_rows_supported(int nchains, const int *nlayers, const det6d_rows_layer *layers);

/* A wide three-layer grouped MLP in ONE launch (csrc/mlp_group.hip): layer 1 from the per-point partial sums exactly as
 * det6d_group_expand, layers 2 and 3 as fp32 MFMA GEMMs on 32-row tiles whose activations stay in LDS (weights streamed
 * from L2 into the MFMA B fragments), then the max-pool: over the nsample rows of a centre with the empty-ball mask
 * (dense rows: idx (B,m,ns), cnt (B*m), ns in {16, 32}) or by class over a compact row list (hdr / crow_p / crow_c; the
 * slice of y must be zeroed: parts of one centre are combined by an integer atomic max).  Bit for bit the sequence
 * det6d_group_expand -> det6d_linear -> det6d_linear(pool).  Widths (c1, c2, c3) in {(128,128,256), (128,256,256),
 * (256,256,512), (256,512,1024)}: det6d_mlp_group3_supported says so.  Replaces the three Conv2d/BN/ReLU + mask +
 * max_pool2d of a radius group (pointnet2_modules.py:462-472) and their two intermediates in memory. */
int det6d_mlp_group3_supported(int c1, int c2, int c3, int ns, int compact);
int det6d_mlp_group3(int rows, const float *p, int ldp, int pcol0, const float *w1, int ldw1, const float *s1, int c1,
                     const float *w2, int ldw2, const float *s2, int c2, const float *w3, int ldw3, const float *s3,
                     int c3, const float *pts, int ldpts, const float *ctr, int ldctr, const int *idx, int n, int m,
                     int ns, const int *cnt, int *hdr, const int *crow_p, const int *crow_c, float *y, int ldy,
                     int col0, det6d_stream_t stream);
/* `hdr` is NOT const: on compact lists the kernel draws its 32-row tiles from the ticket counter hdr[10] and counts its
 * leaving workgroups in hdr[11] (both zero before and after every launch: the last workgroup to leave clears them, and so does
 * det6d_compact_groups when it builds the list).  ONE launch at a time per list: two launches consuming the same list
 * concurrently (two streams, parallel graph branches) would share the counter and skip or repeat tiles — give every launch
 * in flight its own copy of the 16 header words (the row arrays crow_p / crow_c may be shared).  A launch that was aborted
 * leaves the words undefined: rebuild the list or zero hdr[10..11] before reusing it. */

/* First layer of a grouped MLP from per-point partial sums (csrc/expand.hip): with the chain order of gathered rows
 * (feature columns first, relative coordinates last) the feature part P[p][c] = chain_{k >= 3}(row_p[k] * W[k][c]) is one
 * plain det6d_linear over the points (weights with rows 0..2 zeroed, no shift, no activation), and
 *   out[r][0..c1) = act( fma(dz, W[2], fma(dy, W[1], fma(dx, W[0], P[p(r)][pcol0 ..]))) + shift ),  [c1, ldo) zero,
 * for every grouped row r: dense rows (idx (B,m,ns) over n points per scene; hdr == NULL) or a compact row list
 * (hdr / crow_p / crow_c of det6d_compact_groups; rows = capacity, alignment rows zero-filled).  Bit for bit
 * det6d_linear in the gathered modes.  Replaces grouping_operation + cat + the first Conv2d/BN/ReLU of
 * pointnet2_utils.py:449-455 / pointnet2_modules.py:561-568 at 3 FMAs per output instead of 3 + C. */
int det6d_group_expand(int rows, int c1, const float *p, int ldp, int pcol0, const float *w, int ldw,
                       const float *shift, int act, const float *pts, int ldpts, const float *ctr, int ldctr,
                       const int *idx, int n, int m, int ns, const int *hdr, const int *crow_p, const int *crow_c,
                       float *out, int ldo, det6d_stream_t stream);

/* xyz_out[b,j,:] = xyz[b, idx[b*idx_stride + j] + idx_bias, :] (idx_stride >= m: the picks of one sampler inside a
 * layer's concatenated index buffer; idx_bias: `xyz` holds a sub-range of the cloud the indices refer to) and, if rows_out != NULL, the same into columns 0..2 of the next
 * level's rows (B,m,ld_rows) while clearing its padding columns [zero_from, ld_rows). */
int det6d_gather_centres(int b, int n, int m, const float *xyz, const int *idx, int idx_stride, int idx_bias,
                         float *xyz_out, float *rows_out, int ld_rows, int zero_from, det6d_stream_t stream);

/* dst (b*m, 1+ncol) = [batch index, src[b,j,0:ncol]] — the reference's `point_coords`-style tensors
 * (pointnet2_backbone.py:236-240,257-261). */
int det6d_with_batch_index(int b, int m, const float *src, int ld_src, int ncol, float *dst,
                           det6d_stream_t stream);

/* rows_out[b,j,0:ncol] = rows_in[b, idx[b,j], 0:ncol]   (point-major gather; xyz is ncol=3) */
int det6d_gather_rows(int b, int n, int m, int ld_in, int ld_out, int ncol, const float *rows_in,
                      const int *idx, float *rows_out, det6d_stream_t stream);

/* Fused pointwise layer   Y = act(A' * W + shift)  on fp32 MFMA (v_mfma_f32_32x32x2_f32),
 * products accumulated in ascending-k order in one fp32 accumulator per output
 * (bit-identical to the oracle's fmaf chain).
 *
 *   mode A' source:
 *     DET6D_A_ROWS    A'[r][k] = a[r*lda + k]
 *     DET6D_A_GROUPED r = (b, j, s): p = idx[(b*m + j)*ns + s];
 *                     A'[r][k] = a[(b*n + p)*lda + k] - (k < 3 ? ctr[(b*m + j)*ldctr + k] : 0)
 *                     (QueryAndGroup: grouped_xyz -= new_xyz, cat([xyz, feat]);
 *                      pointnet2_utils.py:449-455)
 *   W (K,ldw) row-major with BN folded in, shift (N) = folded BN shift or conv bias (may be NULL),
 *   act: 0 none, 1 ReLU.
 *   pool == 0: y[r*ldy + col0 + c]                      (R rows)
 *   pool == ns: y[(r/ns)*ldy + col0 + c] = max over the ns rows of a group of
 *               act(..) * (cnt[r/ns] > 0)   (mask then max_pool2d, pointnet2_modules.py:465-472)
 */
#define DET6D_A_ROWS 0
#define DET6D_A_GROUPED 1
#define DET6D_A_COMPACT 2   /* A' rows gathered through a compact row list (det6d_compact_groups) */
typedef struct det6d_linear_args {
  int mode;          /* DET6D_A_ROWS / DET6D_A_GROUPED / DET6D_A_COMPACT */
  int rows;          /* R: number of A' rows (= b*m*ns in grouped mode) */
  int k;             /* reduction length (columns of A' actually used) */
  int ncols;         /* N: output channels */
  const float *a; int lda;
  const float *w; int ldw;
  const float *shift;      /* (N) or NULL */
  int act;                 /* 0 none, 1 relu */
  float *y; int ldy; int col0;
  /* grouped mode only */
  int n, m, ns;            /* points per scene, centres per scene, samples per centre */
  const int *idx;          /* (B,m,ns) */
  const float *ctr; int ldctr; /* (B,m,ldctr) centre rows, first 3 columns are xyz */
  /* pooling epilogue */
  int pool;                /* 0, or ns (dense rows), or -1: class pooling over compact rows (hdr, crow_c) */
  const int *cnt;          /* (B*m) hit counts or NULL (no mask) */
  /* compact (ragged) rows, see det6d_compact_groups.  With hdr != NULL the live row count is hdr[0], read on
   * the DEVICE; `rows` is then the capacity (det6d_compact_rows_capacity) the launch is sized for. */
  const int *hdr;
  const int *crow_p;       /* mode COMPACT: point row (scene * n + neighbour) of every compact row; n = B * n */
  const int *crow_c;       /* mode COMPACT / pool -1: centre (scene * m + j) of every compact row (bit 30: empty
                              ball, pooled value 0), -1 = padding */
  int ncols_pad;           /* > ncols: columns [ncols, ncols_pad) of y are written as zeros (the padded width the next
                              layer reads), so callers need no separate fill; 0 = leave them alone */
} det6d_linear_args;
int det6d_linear(const det6d_linear_args *args, det6d_stream_t stream);

/* Compact (ragged) row lists for a grouped MLP.  A ball with cnt < nsample hits is padded by the reference
 * with repetitions of its first cnt hits (ball_query_gpu.cu:75-90,114-129), the MLP is pointwise per row
 * and is followed by a max over the nsample rows (pointnet2_modules.py:462-467): evaluating only the first
 * s = max(smin, 2^ceil(log2 cnt)) slots of every centre gives the same pooled features bit for bit.
 *   cnt (B*m), idx (B,m,ns) as written by the ball queries;  ns, smin powers of two, smin <= ns <= 32;
 *   hdr (det6d_compact_hdr_ints(B*m)) i32: [0] live rows (multiple of 128), [1..6] end of the class regions s = 32,16,8,4,2,1,
 *                 [7] centres, [8] sum of min(cnt, ns), [9] rows before alignment, [10], [11] tile ticket and exit counter of
 *                 the persistent group kernels that consume the list (det6d_mlp_group3: zero between launches; a list must not
 *                 feed two such launches at the same time);
 *   crow_p, crow_c (det6d_compact_rows_capacity(B*m, ns)) i32: point row / centre of every compact row
 *                 (crow_c: bit 30 set for an empty ball, bit 29 for a centre cut into several parts, -1 on
 *                 alignment rows).
 * split = g > 0 (a power of two >= smin): a centre with more than g hits takes ceil(cnt / g) * g rows, cut along their
 * binary digits into parts of descending size (20 = 16 + 4), each a group of its class; consumers combine the parts'
 * maxima with an integer atomic max on the non-negative post-ReLU values, so the pooled rows of MULTI-PART centres must be
 * ZERO before the pooled layer runs; a single-part centre's row is written whole by a plain store.  Centres with <= g hits
 * stay one part of the next power of two >= max(cnt, smin).
 * zero_y != NULL: columns [col0, col0 + width) of the rows of the multi-part centres in the (B*m, ldy) pooled buffer are
 * cleared by this call (all multiples of 4), which saves the separate fill. */
int det6d_compact_rows_capacity(int total_centres, int ns);
int det6d_compact_hdr_ints(int total_centres);   /* ints the hdr buffer must hold (16 header words + scratch) */
int det6d_compact_groups(int b, int n, int m, int ns, int smin, int split, const int *cnt, const int *idx, int *hdr,
                         int *crow_p, int *crow_c, float *zero_y, int ldy, int col0, int width, det6d_stream_t stream);

/* det6d_compact_groups for both radius groups (a, b) of one SA layer in one pair of launches */
int det6d_compact_groups_pair(int b, int n, int m, int smin, int split, int ns_a, const int *cnt_a, const int *idx_a,
                              int *hdr_a, int *crow_p_a, int *crow_c_a, int col0_a, int width_a, int ns_b,
                              const int *cnt_b, const int *idx_b, int *hdr_b, int *crow_p_b, int *crow_c_b, int col0_b,
                              int width_b, float *zero_y, int ldy, det6d_stream_t stream);

/* The grid query as the compact-row engine uses it, and the list builder behind it.
 * det6d_ball_query_pair_grid_lists: same hits and counts as det6d_ball_query_pair_grid, but (1) index rows are written only as
 * far as the list builder reads them — the slots below the next power of two >= max(cnt, 4), the reference's cyclic padding
 * (ball_query_gpu.cu:75-90,114-129) included; the rest of an idx row is left untouched — and (2) every workgroup (256
 * consecutive centres of a scene) leaves the part counts of its centres in the scratch part of hdr_a / hdr_b, so that
 * det6d_compact_groups_pair_counted only places rows (one launch less per SA layer).  m must be a multiple of 256, smin <= 4,
 * ns_a / ns_b in {4, 8, 16, 32}; smin / split as for det6d_compact_groups_pair, to which the pair is equivalent. */
int det6d_ball_query_pair_grid_lists(int b, int n, int m, float rin_a, float rout_a, int ns_a, float rin_b, float rout_b,
                                     int ns_b, const float *new_xyz, const float *xyz, void *workspace, int *cnt_a, int *idx_a,
                                     int *cnt_b, int *idx_b, int smin, int split, int *hdr_a, int *hdr_b,
                                     det6d_stream_t stream);
int det6d_compact_groups_pair_counted(int b, int n, int m, int smin, int split, int ns_a, const int *cnt_a, const int *idx_a,
                                      int *hdr_a, int *crow_p_a, int *crow_c_a, int col0_a, int width_a, int ns_b,
                                      const int *cnt_b, const int *idx_b, int *hdr_b, int *crow_p_b, int *crow_c_b, int col0_b,
                                      int width_b, float *zero_y, int ldy, det6d_stream_t stream);

/* The three pointwise layers of a grouped MLP in one launch, nsample 16 or 32: identical, bit for bit, to
 *   det6d_linear(GROUPED, W1, ReLU) -> det6d_linear(ROWS, W2, ReLU) -> det6d_linear(ROWS, W3, ReLU, pool = ns, cnt)
 * but the (rows x c1), (rows x c2) intermediates never leave the CU (csrc/mlp_chain.hip).  Supported shapes:
 *   narrow: row width lda <= 8, c1, c2 <= 32, c3 <= 64 (register kernel for lda = 4 with (16,16,32) / (32,32,64),
 *           LDS kernel otherwise);
 *   wide:   lda = 68, c1 = 64, c2 = 64 or 96, c3 = 128, 16-byte aligned weights / shifts, m even when ns = 16.
 * Anything else returns DET6D_EINVAL (callers fall back to three det6d_linear calls).
 * y[(r / ns) * ldy + col0 + c], r over b*m*ns rows. */
int det6d_mlp_chain3(int rows, int n, int m, int ns, const float *a, int lda, const int *idx,
                     const float *ctr, int ldctr, const int *cnt, const float *w1, int ldw1, const float *s1,
                     int c1, const float *w2, int ldw2, const float *s2, int c2, const float *w3, int ldw3,
                     const float *s3, int c3, float *y, int ldy, int col0, det6d_stream_t stream);

/* det6d_mlp_chain3 over a compact row list (register kernels only: lda = 4 with (16,16,32) / (32,32,64), or the
 * wide shapes); y[centre * ldy + col0 + c] for every centre of the list. */
int det6d_mlp_chain3_compact(int capacity, const int *hdr, const int *crow_p, const int *crow_c, const float *a,
                             int lda, const float *ctr, int ldctr, const float *w1, int ldw1, const float *s1, int c1,
                             const float *w2, int ldw2, const float *s2, int c2, const float *w3, int ldw3,
                             const float *s3, int c3, float *y, int ldy, int col0, det6d_stream_t stream);

/* Mask + max-pool of a grouped MLP's last layer for ANY nsample (pointnet2_modules.py:465-472: new_features *= (idx_cnt > 0);
 * F.max_pool2d(kernel_size=[1, nsample])):  y[r*ldy + col0 + c] = cnt[r] > 0 ? max_s x[(r*ns + s)*ldx + c] : 0  for r < groups,
 * c < ncols (cnt == NULL: no mask).  The fused GEMM epilogues pool nsample in {8, 16, 32} themselves; this entry is the
 * fallback for every other nsample. */
int det6d_group_maxpool(int groups, int ns, int ncols, const float *x, int ldx, const int *cnt, float *y, int ldy, int col0,
                        det6d_stream_t stream);

/* s-fps weights: w[i] = sigmoid(score[i]) ** gamma  (pointnet2_modules.py:415-419) */
int det6d_sigmoid_pow(int count, const float *scores, float gamma, float *weights,
                      det6d_stream_t stream);

/* vote_xyz = cand_xyz + clamp(offset, -range, +range)
 * (point_head_box6d_vote.py:816-821).  off (R,ldo) first 3 cols; cand (R,ldc) first 3 cols;
 * out vote (R,ldv) first 3 cols; also writes the clamped offsets to off_out (R,3) if non-NULL */
int det6d_vote_points(int rows, const float *off, int ldo, const float *cand, int ldc,
                      float rx, float ry, float rz, float *vote, int ldv, float *off_out,
                      det6d_stream_t stream);

/* PointBinResidual6DCoder.decode_torch with use_mean_size = False
 * (box_coder_utils.py:589-603,622-680,723-737).
 *   code (R,ldcode) = 6 offsets + nbin yaw logits + nbin yaw residuals + (ground_aware ? 2 : 1);
 *   pts (R,ldp) first 3 cols; boxes (R,9) = [x,y,z,dx,dy,dz,rz,ry,rx]. */
int det6d_decode_boxes(int rows, int nbin, int ground_aware, int minus, float threshold_rad,
                       float factor_rad, const float *code, int ldcode, const float *pts, int ldp,
                       float *boxes, det6d_stream_t stream);

/* Per-scene post-processing (Detector3DTemplate.post_processing eval branch +
 * class_agnostic_nms + nms_gpu: detector3d_template.py:178-284, model_nms_utils.py:6-25,
 * iou3d_nms_utils.py:84-99), one workgroup per scene, no host sync:
 *   cls (B*P,ncls) logits; boxes (B*P,9); per scene: sigmoid, max over classes, score >= thr,
 *   stable descending sort, top pre_max, rotated NMS (dims 0..6) at nms_thr, first post_max.
 * Outputs (device): out_boxes (B,post_max,9), out_scores (B,post_max), out_labels (B,post_max)
 * i32 1-based, out_index (B,post_max) i32 = index of the kept box inside its scene, out_count (B).
 * `workspace`: det6d_postprocess_workspace_bytes(B) bytes of 16-byte-aligned device scratch
 * (scores, order, sorted boxes, suppression matrix), owned by the caller like every other buffer.
 * P <= 1024 (the suppression matrix is staged in LDS for the greedy scan). Equal scores are ordered by ascending original index (the reference uses an
 * unstable torch sort there). */
int64_t det6d_postprocess_workspace_bytes(int b);
int det6d_postprocess(int b, int p, int ncls, const float *cls, const float *boxes, float score_thr,
                      int pre_max, int post_max, float nms_thr, void *workspace, float *out_boxes,
                      float *out_scores, int *out_labels, int *out_index, int *out_count,
                      det6d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DET6D_OPS_H */
