// slope.hip — SlopeAug geometry on the device.
//
//   random_global_make_slope   core/pcdet/datasets/augmentor/augmentor_utils.py:622-694
//       (the non-smooth branch :670-694; the smooth branch is two calls of it, driven by the host)
//   boxes3d_to_corners_3d      core/pcdet/utils/box_utils.py:57-71   (9-D boxes, euler 'zyx')
//
// The random pivot / angle are a handful of host scalars (drawn with the same np.random calls as the
// reference by the Python mirror); what runs here is the per-point and per-box work: everything on
// the far side of the pivot line is rotated about the pivot by the rotation vector, boxes get their
// pitch / roll increments and angles wrapped to [-pi, pi).
// Arithmetic follows NumPy's: float32 points are updated in place through float64 temporaries
// (`p -= pivot`, `p = p @ R^T`, `p += pivot`, each rounded back to float32), boxes stay float64.
// Dots are ascending fma chains (NumPy: BLAS) -> documented tolerance 1e-6 instead of bit parity.
#include "common.h"

namespace {

struct SlopeArgs {
  double pivot[3];
  double rot[9];      // row-major R (Rotation.from_rotvec(angle).as_matrix())
  double k, y0, x0;   // pivot line  y = k (x - x0) + y0
  double side;        // np.sign(k * (0 - x0) + y0 - 0): the sensor's side of the line
  double d_pitch, d_roll;
};

__device__ __forceinline__ double sgn(double v) { return (v > 0.0) - (v < 0.0); }

__device__ __forceinline__ void rotate3(const double *R, double x, double y, double z, double *o) {
#pragma unroll
  for (int j = 0; j < 3; ++j) o[j] = fma(z, R[3 * j + 2], fma(y, R[3 * j + 1], x * R[3 * j]));
}

__global__ __launch_bounds__(256) void slope_points_kernel(int n, float *points, int ld, const SlopeArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float *p = points + (size_t)i * ld;
  const double x = p[0], y = p[1];
  if (sgn(a.k * (x - a.x0) + a.y0 - y) == a.side) return;
  const float fx = (float)(x - a.pivot[0]), fy = (float)(y - a.pivot[1]), fz = (float)((double)p[2] - a.pivot[2]);
  double r[3];
  rotate3(a.rot, fx, fy, fz, r);
  p[0] = (float)((double)(float)r[0] + a.pivot[0]);
  p[1] = (float)((double)(float)r[1] + a.pivot[1]);
  p[2] = (float)((double)(float)r[2] + a.pivot[2]);
}

// common_utils.limit_period(v, 0.5, 2 pi).  The reference routes NumPy input through torch float32
// (check_numpy_to_torch: .float()), so every step rounds to float and the wrapped angle is a float32 value.
__device__ __forceinline__ double wrap_pi(double v) {
  const float period = 6.283185307179586f, x = (float)v;
  const float q = floorf(x / period + 0.5f);
  const float t = q * period;
  return (double)(x - t);
}

__global__ __launch_bounds__(64) void slope_boxes_kernel(int m, double *boxes, const SlopeArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double *b = boxes + (size_t)i * 9;
  if (sgn(a.k * (b[0] - a.x0) + a.y0 - b[1]) != a.side) {
    double r[3];
    rotate3(a.rot, b[0] - a.pivot[0], b[1] - a.pivot[1], b[2] - a.pivot[2], r);
    b[0] = r[0] + a.pivot[0]; b[1] = r[1] + a.pivot[1]; b[2] = r[2] + a.pivot[2];
    b[7] += a.d_pitch;
    b[8] += a.d_roll;
  }
  b[6] = wrap_pi(b[6]); b[7] = wrap_pi(b[7]); b[8] = wrap_pi(b[8]);
}

// corners of 9-D boxes [x,y,z,dx,dy,dz,rz,ry,rx]: R = Rx(rx) Ry(ry) Rz(rz)  (scipy from_euler('zyx', [rz,ry,rx]): extrinsic z, y, x)
__global__ __launch_bounds__(64) void boxes9_corners_kernel(int m, const double *boxes, double *corners) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const double *b = boxes + (size_t)i * 9;
  const double cz = cos(b[6]), sz = sin(b[6]), cy = cos(b[7]), sy = sin(b[7]), cx = cos(b[8]), sx = sin(b[8]);
  const double R[9] = {cy * cz, -cy * sz, sy,
                       cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy,
                       sx * sz - cx * sy * cz, sx * cz + cx * sy * sz, cx * cy};
  for (int k = 0; k < 8; ++k) {
    // template of the reference: x = +,+,-,-,+,+,-,-  y = +,-,-,+,+,-,-,+  z = -,-,-,-,+,+,+,+
    const double lx = ((k & 3) < 2 ? b[3] : -b[3]) * 0.5;
    const double ly = (((k & 3) == 0 || (k & 3) == 3) ? b[4] : -b[4]) * 0.5;
    const double lz = (k < 4 ? -b[5] : b[5]) * 0.5;
    double r[3];
    rotate3(R, lx, ly, lz, r);
    double *o = corners + ((size_t)i * 8 + k) * 3;
    o[0] = r[0] + b[0]; o[1] = r[1] + b[1]; o[2] = r[2] + b[2];
  }
}

}  // namespace

DET6D_API int det6d_make_slope(int n_points, float *points, int ld, int n_boxes, double *boxes9, const double *params,
                               det6d_stream_t stream) {
  if (n_points < 0 || n_boxes < 0 || !params || (n_points > 0 && (!points || ld < 3)) || (n_boxes > 0 && !boxes9))
    return DET6D_EINVAL;
  SlopeArgs a;
  for (int j = 0; j < 3; ++j) a.pivot[j] = params[j];
  for (int j = 0; j < 9; ++j) a.rot[j] = params[3 + j];
  a.k = params[12]; a.x0 = params[0]; a.y0 = params[1]; a.side = params[13]; a.d_pitch = params[14]; a.d_roll = params[15];
  hipStream_t s = (hipStream_t)stream;
  if (n_points > 0)
    hipLaunchKernelGGL(slope_points_kernel, dim3(det6d_divup(n_points, 256)), dim3(256), 0, s, n_points, points, ld, a);
  if (n_boxes > 0)
    hipLaunchKernelGGL(slope_boxes_kernel, dim3(det6d_divup(n_boxes, 64)), dim3(64), 0, s, n_boxes, boxes9, a);
  return det6d_check_launch("det6d_make_slope");
}

DET6D_API int det6d_boxes9_corners(int n_boxes, const double *boxes9, double *corners, det6d_stream_t stream) {
  if (n_boxes < 0) return DET6D_EINVAL;
  if (n_boxes == 0) return DET6D_OK;
  if (!boxes9 || !corners) return DET6D_EINVAL;
  hipLaunchKernelGGL(boxes9_corners_kernel, dim3(det6d_divup(n_boxes, 64)), dim3(64), 0, (hipStream_t)stream, n_boxes,
                     boxes9, corners);
  return det6d_check_launch("det6d_boxes9_corners");
}
