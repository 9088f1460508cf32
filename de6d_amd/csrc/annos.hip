// annos.hip — the output consumer of the hot path on the device.
//
// Turns the detections of a whole batch (LiDAR frame, 7 or 9 columns) into the KITTI annotation
// fields in one launch, so that the host receives ONE packed array per batch instead of pulling three
// tensors per scene and converting them in NumPy:
//   box_utils.boxes3d_lidar_to_kitti_camera        core/pcdet/utils/box_utils.py:196-212
//   box_utils.boxes3d_to_corners3d_kitti_camera    box_utils.py:215-258
//   box_utils.boxes3d_kitti_camera_to_imageboxes   box_utils.py:261-281
//   Calibration.lidar_to_rect / rect_to_img        core/pcdet/utils/calibration_kitti.py:64-83
//   alpha                                          core/pcdet/datasets/kitti/kitti_dataset.py:319
// float32 throughout, like the reference's NumPy path (float32 boxes and calibration matrices).
// The reference's dot products go through BLAS sgemm, whose summation order is not defined; here
// every dot is an ascending fma chain, hence the documented 1e-4 tolerance instead of bit parity.
#include "common.h"

namespace {

constexpr int kCalibFloats = 28;  // M (4x3) | P2 (3x4) | image h, w | 2 pad
constexpr int kAnnoFloats = 12;   // camera box x,y,z,l,h,w,r | bbox x1,y1,x2,y2 | alpha

__global__ __launch_bounds__(256) void kitti_annos_kernel(int total, const float *boxes, int ld, const int *scene_of,
                                                         const float *calib, float *out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const float *b = boxes + (size_t)t * ld;
  const float *c = calib + (size_t)scene_of[t] * kCalibFloats;
  const float *M = c, *P = c + 12;
  const float img_h = c[24], img_w = c[25];
  const float x = b[0], y = b[1], l = b[3], w = b[4], h = b[5], heading = b[6];
  const float z = b[2] - h / 2.f;                      // box centre -> bottom centre
  float cam[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) cam[j] = D6_FMA(z, M[6 + j], D6_FMA(y, M[3 + j], x * M[j])) + M[9 + j];
  const float ry = -heading - 1.57079632679489661923f;
  float sn, cs;
  d6_sincosf(ry, &sn, &cs);
  float u0 = 3.0e38f, v0 = 3.0e38f, u1 = -3.0e38f, v1 = -3.0e38f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // corner order of the reference: x = +,+,-,-,+,+,-,-   z = +,-,-,+,+,-,-,+   y = 0 (bottom) x4, -h (top) x4
    const float xc = ((k & 3) < 2 ? l : -l) / 2.f;
    const float zc = (((k & 3) == 0 || (k & 3) == 3) ? w : -w) / 2.f;
    const float yc = k < 4 ? 0.f : -h;
    const float px = cam[0] + D6_FMA(zc, sn, xc * cs);
    const float py = cam[1] + yc;
    const float pz = cam[2] + D6_FMA(zc, cs, -xc * sn);
    const float hu = D6_FMA(pz, P[2], D6_FMA(py, P[1], px * P[0])) + P[3];
    const float hv = D6_FMA(pz, P[6], D6_FMA(py, P[5], px * P[4])) + P[7];
    const float u = hu / pz, v = hv / pz;              // the reference divides by the rect z, not by the homogeneous w
    u0 = d6_fminf(u0, u); u1 = d6_fmaxf(u1, u);
    v0 = d6_fminf(v0, v); v1 = d6_fmaxf(v1, v);
  }
  if (img_w > 0.f) {
    u0 = d6_fminf(d6_fmaxf(u0, 0.f), img_w - 1.f); u1 = d6_fminf(d6_fmaxf(u1, 0.f), img_w - 1.f);
    v0 = d6_fminf(d6_fmaxf(v0, 0.f), img_h - 1.f); v1 = d6_fminf(d6_fmaxf(v1, 0.f), img_h - 1.f);
  }
  float *o = out + (size_t)t * kAnnoFloats;
  o[0] = cam[0]; o[1] = cam[1]; o[2] = cam[2];
  o[3] = l; o[4] = h; o[5] = w; o[6] = ry;
  o[7] = u0; o[8] = v0; o[9] = u1; o[10] = v1;
  o[11] = -d6_atan2f(-y, x) + ry;
}

}  // namespace

DET6D_API int det6d_kitti_annos(int total, const float *boxes, int ld, const int *scene_of, const float *calib,
                                float *annos_out, det6d_stream_t stream) {
  if (total < 0 || ld < 7) return DET6D_EINVAL;
  if (total == 0) return DET6D_OK;
  if (!boxes || !scene_of || !calib || !annos_out) return DET6D_EINVAL;
  hipLaunchKernelGGL(kitti_annos_kernel, dim3(det6d_divup(total, 256)), dim3(256), 0, (hipStream_t)stream, total, boxes,
                     ld, scene_of, calib, annos_out);
  return det6d_check_launch("det6d_kitti_annos");
}
