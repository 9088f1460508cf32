// points.hip — the bandwidth-bound index/elementwise ops of the path (gather, group, 3-NN,
// interpolation, point packing, vote clamp, box decode).  One thread per output element,
// grid-stride, coalesced on the output side; all of them are HBM/L2-bound byte movers.
//
// Replaces core/pcdet/ops/pointnet2/pointnet2_batch/src/{sampling_gpu.cu:16-90,
// group_points_gpu.cu:14-92, interpolate_gpu.cu:16-170}.
#include "common.h"

namespace {

constexpr int kBlock = 256;

inline dim3 grid_for(int64_t total) {
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond ~32 blocks per CU
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks);
}

#define GRID_STRIDE(i, total)                                                       \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total);     \
       i += (int64_t)gridDim.x * blockDim.x)

// out[b,c,j] = points[b,c,idx[b,j]]
__global__ void gather_points_kernel(int64_t total, int c, int n, int m, const float *__restrict__ points,
                                     const int *__restrict__ idx, float *__restrict__ out) {
  GRID_STRIDE(i, total) {
    const int j = (int)(i % m);
    const int64_t bc = i / m;
    const int64_t bi = bc / c;
    out[i] = points[bc * n + idx[bi * m + j]];
  }
}

__global__ void gather_points_grad_kernel(int64_t total, int c, int n, int m,
                                          const float *__restrict__ grad_out,
                                          const int *__restrict__ idx, float *__restrict__ grad_points) {
  GRID_STRIDE(i, total) {
    const int j = (int)(i % m);
    const int64_t bc = i / m;
    const int64_t bi = bc / c;
    atomicAdd(grad_points + bc * n + idx[bi * m + j], grad_out[i]);
  }
}

// out[b,c,p,s] = points[b,c,idx[b,p,s]]
__global__ void group_points_kernel(int64_t total, int c, int n, int64_t ms,
                                    const float *__restrict__ points, const int *__restrict__ idx,
                                    float *__restrict__ out) {
  GRID_STRIDE(i, total) {
    const int64_t ps = i % ms;
    const int64_t bc = i / ms;
    const int64_t bi = bc / c;
    out[i] = points[bc * n + idx[bi * ms + ps]];
  }
}

__global__ void group_points_grad_kernel(int64_t total, int c, int n, int64_t ms,
                                         const float *__restrict__ grad_out,
                                         const int *__restrict__ idx, float *__restrict__ grad_points) {
  GRID_STRIDE(i, total) {
    const int64_t ps = i % ms;
    const int64_t bc = i / ms;
    const int64_t bi = bc / c;
    atomicAdd(grad_points + bc * n + idx[bi * ms + ps], grad_out[i]);
  }
}

// interpolate_gpu.cu:16-59: double running bests initialised to 1e40, strict '<' cascade.
__global__ void three_nn_kernel(int n, int m, const float *__restrict__ unknown,
                                const float *__restrict__ known, float *__restrict__ dist2,
                                int *__restrict__ idx) {
  __shared__ float tile[3 * 512];
  const int bs = blockIdx.y;
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = pt < n;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (live) {
    const float *u = unknown + ((size_t)bs * n + pt) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  const float *kn = known + (size_t)bs * m * 3;
  // The reference keeps the running bests as double 1e40 and compares float candidates against them; a float candidate
  // converts to double exactly, so float bests initialised to +inf select the same points (d < 1e40 <=> d < inf for every
  // float d; an untouched best prints as (float)1e40 = inf either way).  b1 <= b2 <= b3 always holds, which turns the
  // reference's if / else-if cascade into three independent compares + selects (no divergence, no f64 ops).
  float best1 = __builtin_inff(), best2 = __builtin_inff(), best3 = __builtin_inff();
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int k0 = 0; k0 < m; k0 += 512) {
    const int cnt = min(512, m - k0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * 3; t += blockDim.x) tile[t] = kn[(size_t)k0 * 3 + t];
    __syncthreads();
    if (live) {
#pragma unroll 4
      for (int kk = 0; kk < cnt; ++kk) {
        const float d = d6_sqdist(ux - tile[kk * 3 + 0], uy - tile[kk * 3 + 1], uz - tile[kk * 3 + 2]);
        const int k = k0 + kk;
        const bool c1 = d < best1, c2 = d < best2, c3 = d < best3;
        best3 = c2 ? best2 : (c3 ? d : best3);  besti3 = c2 ? besti2 : (c3 ? k : besti3);
        best2 = c1 ? best1 : (c2 ? d : best2);  besti2 = c1 ? besti1 : (c2 ? k : besti2);
        best1 = c1 ? d : best1;                 besti1 = c1 ? k : besti1;
      }
    }
  }
  if (live) {
    float *od = dist2 + ((size_t)bs * n + pt) * 3;
    int *oi = idx + ((size_t)bs * n + pt) * 3;
    od[0] = best1; od[1] = best2; od[2] = best3;
    oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
  }
}

// out[b,c,i] = fma(w2,p2, fma(w0,p0, w1*p1))   (contraction order: oracle/det6d_oracle.c)
__global__ void three_interpolate_kernel(int64_t total, int c, int m, int n,
                                         const float *__restrict__ points, const int *__restrict__ idx,
                                         const float *__restrict__ weight, float *__restrict__ out) {
  GRID_STRIDE(i, total) {
    const int pt = (int)(i % n);
    const int64_t bc = i / n;
    const int64_t bi = bc / c;
    const float *w = weight + (bi * n + pt) * 3;
    const int *id = idx + (bi * n + pt) * 3;
    const float *p = points + bc * m;
    out[i] = D6_FMA(w[2], p[id[2]], D6_FMA(w[0], p[id[0]], w[1] * p[id[1]]));
  }
}

// LDS form for feature maps whose channel rows fit the CU's LDS (m <= 16384): a workgroup stages CH channel rows of one scene
// (coalesced, once) and every thread interpolates its points for those channels from LDS.  The gathers of the plain kernel are
// 64 random 4-byte reads of a 4 m-byte row per wave instruction (~40 cache lines each): it ran at 0.85 TB/s of algorithmic
// bytes for (8, 64, 4096) -> 16384 points; from LDS a gather costs a few bank-conflict cycles.  Same arithmetic per element.
template <int CH>
__global__ __launch_bounds__(256) void three_interpolate_lds_kernel(int c, int m, int n, int pts_per_wg, const float *__restrict__ points,
                                                                    const int *__restrict__ idx, const float *__restrict__ weight,
                                                                    float *__restrict__ out) {
  extern __shared__ float rows[];                  // CH x m
  const int64_t bi = blockIdx.z;
  const int c0 = blockIdx.y * CH, nch = min(CH, c - c0);
  const float *src = points + (bi * c + c0) * m;
  const int total = nch * m;
  if (((uintptr_t)src & 15) == 0 && (total & 3) == 0) {
    for (int t = 4 * threadIdx.x; t < total; t += 4 * 256) *reinterpret_cast<float4 *>(rows + t) = *reinterpret_cast<const float4 *>(src + t);
  } else {
    for (int t = threadIdx.x; t < total; t += 256) rows[t] = src[t];
  }
  __syncthreads();
  const int p0 = blockIdx.x * pts_per_wg, p1 = min(p0 + pts_per_wg, n);
  for (int pt = p0 + threadIdx.x; pt < p1; pt += 256) {
    const float *w = weight + (bi * n + pt) * 3;
    const int *id = idx + (bi * n + pt) * 3;
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    float *o = out + (bi * c + c0) * n + pt;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
      if (ch < nch) o[(size_t)ch * n] = D6_FMA(w2, rows[ch * m + i2], D6_FMA(w0, rows[ch * m + i0], w1 * rows[ch * m + i1]));
  }
}

__global__ void three_interpolate_grad_kernel(int64_t total, int c, int n, int m,
                                              const float *__restrict__ grad_out,
                                              const int *__restrict__ idx,
                                              const float *__restrict__ weight,
                                              float *__restrict__ grad_points) {
  GRID_STRIDE(i, total) {
    const int pt = (int)(i % n);
    const int64_t bc = i / n;
    const int64_t bi = bc / c;
    const float *w = weight + (bi * n + pt) * 3;
    const int *id = idx + (bi * n + pt) * 3;
    float *gp = grad_points + bc * m;
    const float g = grad_out[i];
    atomicAdd(gp + id[0], g * w[0]);
    atomicAdd(gp + id[1], g * w[1]);
    atomicAdd(gp + id[2], g * w[2]);
  }
}

__global__ void pack_points_kernel(int64_t total, int width, int ld, const float *__restrict__ points,
                                   float *__restrict__ rows, float *__restrict__ xyz_out) {
  GRID_STRIDE(i, total) {
    const int c = (int)(i % ld);
    const int64_t r = i / ld;
    const float v = c < width ? points[r * (width + 1) + 1 + c] : 0.f;
    rows[i] = v;
    if (xyz_out && c < 3) xyz_out[r * 3 + c] = v;
  }
}

// centres of a layer: xyz_out[b,j,:] = xyz[b,idx[b,j],:]; optionally the same three columns into the
// next level's row buffer, whose padding columns [zero_from, ld_rows) are cleared here as well
__global__ void gather_centres_kernel(int64_t total, int n, int m, int idx_stride, int idx_bias, int ld_rows, int zero_from,
                                      const float *__restrict__ xyz, const int *__restrict__ idx,
                                      float *__restrict__ xyz_out, float *__restrict__ rows_out) {
  const int per = 3 + (rows_out ? ld_rows - zero_from : 0);
  GRID_STRIDE(i, total) {
    const int c = (int)(i % per);
    const int64_t bj = i / per;
    if (c < 3) {
      const int64_t bi = bj / m;
      const float v = xyz[(bi * n + idx[bi * idx_stride + (bj - bi * m)] + idx_bias) * 3 + c];
      xyz_out[bj * 3 + c] = v;
      if (rows_out) rows_out[bj * ld_rows + c] = v;
    } else {
      rows_out[bj * ld_rows + zero_from + (c - 3)] = 0.f;
    }
  }
}

// dst[b*m + j, :] = [b, src[b, j, 0:ncol]]   (the reference's (N, 1 + 3) "batch index + xyz" tensors)
__global__ void with_batch_index_kernel(int64_t total, int m, int ld_src, int ncol,
                                        const float *__restrict__ src, float *__restrict__ dst) {
  GRID_STRIDE(i, total) {
    const int c = (int)(i % (ncol + 1));
    const int64_t bj = i / (ncol + 1);
    dst[i] = c == 0 ? (float)(bj / m) : src[bj * ld_src + c - 1];
  }
}

// mask + max over the ns rows of a group, one thread per (group, column): consecutive threads read consecutive columns
__global__ void group_maxpool_kernel(int64_t total, int ns, int ncols, const float *__restrict__ x, int ldx,
                                     const int *__restrict__ cnt, float *__restrict__ y, int ldy, int col0) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / ncols;
    const int c = (int)(t - r * ncols);
    const float *p = x + (size_t)r * ns * ldx + c;
    float v = p[0];
    for (int s = 1; s < ns; ++s) v = d6_vmax(v, p[(size_t)s * ldx]);
    if (cnt && cnt[r] <= 0) v = 0.f;
    y[(size_t)r * ldy + col0 + c] = v;
  }
}

__global__ void gather_rows_kernel(int64_t total, int n, int m, int ld_in, int ld_out, int ncol,
                                   const float *__restrict__ rows_in, const int *__restrict__ idx,
                                   float *__restrict__ rows_out) {
  GRID_STRIDE(i, total) {
    const int c = (int)(i % ncol);
    const int64_t bj = i / ncol;
    const int64_t bi = bj / m;
    rows_out[bj * ld_out + c] = rows_in[(bi * n + idx[bj]) * ld_in + c];
  }
}

__global__ void sigmoid_pow_kernel(int64_t total, const float *__restrict__ s, float gamma,
                                   float *__restrict__ w) {
  GRID_STRIDE(i, total) w[i] = d6_sigmoid_powf(s[i], gamma);
}

__global__ void vote_points_kernel(int64_t total, const float *__restrict__ off, int ldo,
                                   const float *__restrict__ cand, int ldc, float rx, float ry, float rz,
                                   float *__restrict__ vote, int ldv, float *__restrict__ off_out) {
  GRID_STRIDE(i, total) {
    const int c = (int)(i % 3);
    const int64_t r = i / 3;
    const float R = c == 0 ? rx : (c == 1 ? ry : rz);
    float o = off[r * ldo + c];
    o = o > -R ? o : -R;
    o = o < R ? o : R;
    if (off_out) off_out[r * 3 + c] = o;
    vote[r * ldv + c] = cand[r * ldc + c] + o;
  }
}

// box_coder_utils.py:589-603,622-680 (PointBinResidual6DCoder.decode_torch, use_mean_size=False)
__global__ void decode_boxes_kernel(int rows, int nbin, int ground_aware, int minus, float thr,
                                    float fac, float per_bin, const float *__restrict__ code, int ldcode,
                                    const float *__restrict__ pts, int ldp, float *__restrict__ boxes) {
  GRID_STRIDE(r, rows) {
    const float *c = code + r * ldcode;
    const float *p = pts + r * ldp;
    float *o = boxes + r * 9;
    o[0] = c[0] + p[0];
    o[1] = c[1] + p[1];
    o[2] = c[2] + p[2];
    o[3] = d6_expf(c[3]);
    o[4] = d6_expf(c[4]);
    o[5] = d6_expf(c[5]);
    const float *bin = c + 6, *res = c + 6 + nbin, *gr = c + 6 + 2 * nbin;
    int am = 0;
    float bv = bin[0];
    for (int i = 1; i < nbin; ++i) {
      const float v = bin[i];
      if (v > bv) { bv = v; am = i; }
    }
    o[6] = ((float)am + res[am]) * per_bin;
    if (ground_aware) {
      const bool no_pitch = d6_sigmoidf(gr[0]) < 0.5f;
      float pitch = minus ? gr[1] * fac : (-thr) - gr[1] * fac;
      if (no_pitch) pitch = 0.f;
      o[7] = pitch;
    } else {
      o[7] = gr[0];
    }
    o[8] = 0.f;
  }
}

}  // namespace

#define S(x) ((hipStream_t)(x))

DET6D_API int det6d_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                                  float *out, det6d_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || !points || !idx || !out) return DET6D_EINVAL;
  const int64_t total = (int64_t)b * c * npoints;
  if (total == 0) return DET6D_OK;
  hipLaunchKernelGGL(gather_points_kernel, grid_for(total), dim3(kBlock), 0, S(stream), total, c, n,
                     npoints, points, idx, out);
  return det6d_check_launch("det6d_gather_points");
}

DET6D_API int det6d_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                       const int *idx, float *grad_points, det6d_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || !grad_out || !idx || !grad_points) return DET6D_EINVAL;
  const int64_t total = (int64_t)b * c * npoints;
  if (total == 0) return DET6D_OK;
  hipLaunchKernelGGL(gather_points_grad_kernel, grid_for(total), dim3(kBlock), 0, S(stream), total, c, n,
                     npoints, grad_out, idx, grad_points);
  return det6d_check_launch("det6d_gather_points_grad");
}

DET6D_API int det6d_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                                 const int *idx, float *out, det6d_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0 || !points || !idx || !out) return DET6D_EINVAL;
  const int64_t ms = (int64_t)npoints * nsample;
  const int64_t total = (int64_t)b * c * ms;
  if (total == 0) return DET6D_OK;
  hipLaunchKernelGGL(group_points_kernel, grid_for(total), dim3(kBlock), 0, S(stream), total, c, n, ms,
                     points, idx, out);
  return det6d_check_launch("det6d_group_points");
}

DET6D_API int det6d_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                      const float *grad_out, const int *idx, float *grad_points,
                                      det6d_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0 || !grad_out || !idx || !grad_points)
    return DET6D_EINVAL;
  const int64_t ms = (int64_t)npoints * nsample;
  const int64_t total = (int64_t)b * c * ms;
  if (total == 0) return DET6D_OK;
  hipLaunchKernelGGL(group_points_grad_kernel, grid_for(total), dim3(kBlock), 0, S(stream), total, c, n,
                     ms, grad_out, idx, grad_points);
  return det6d_check_launch("det6d_group_points_grad");
}

DET6D_API int det6d_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                             int *idx, det6d_stream_t stream) {
  if (b < 0 || n < 0 || m < 0 || !unknown || !known || !dist2 || !idx) return DET6D_EINVAL;
  if (b == 0 || n == 0) return DET6D_OK;
  hipLaunchKernelGGL(three_nn_kernel, dim3(det6d_divup(n, kBlock), b), dim3(kBlock), 0, S(stream), n, m,
                     unknown, known, dist2, idx);
  return det6d_check_launch("det6d_three_nn");
}

DET6D_API int det6d_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                      const float *weight, float *out, det6d_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || m < 0 || !points || !idx || !weight || !out) return DET6D_EINVAL;
  const int64_t total = (int64_t)b * c * n;
  if (total == 0) return DET6D_OK;
  // channel rows through LDS: as many (4, 2 or 1) as fit 64 KB (two workgroups per CU): m <= 4096 / 8192 / 16384.  Like the
  // reference's kernel (interpolate_gpu.cu:84-104) it trusts idx to lie in [0, m).
  const int ch = m <= 0 ? 0 : (size_t)m * 16 <= 64 * 1024 ? 4 : (size_t)m * 8 <= 64 * 1024 ? 2 : (size_t)m * 4 <= 64 * 1024 ? 1 : 0;
  if (ch && n >= 1024 && b <= 65535 && det6d_divup(c, ch) <= 65535) {
    int splits = det6d_divup(1024, b * det6d_divup(c, ch));          // >= ~1024 workgroups in all
    if (splits < 1) splits = 1;
    int pts = det6d_divup(det6d_divup(n, splits), 256) * 256;
    if (pts < 256) pts = 256;
    const size_t lds = (size_t)m * ch * 4;
    const dim3 grid(det6d_divup(n, pts), det6d_divup(c, ch), b);
#define D6_INTERP_LDS(CH)                                                                                                     \
  do {                                                                                                                        \
    DET6D_MAX_DYNAMIC_LDS(three_interpolate_lds_kernel<CH>, 64 * 1024);                                                         \
    hipLaunchKernelGGL(three_interpolate_lds_kernel<CH>, grid, dim3(256), lds, S(stream), c, m, n, pts, points, idx, weight, out); \
  } while (0)
    if (ch == 4) D6_INTERP_LDS(4);
    else if (ch == 2) D6_INTERP_LDS(2);
    else D6_INTERP_LDS(1);
#undef D6_INTERP_LDS
    return det6d_check_launch("det6d_three_interpolate");
  }
  hipLaunchKernelGGL(three_interpolate_kernel, grid_for(total), dim3(kBlock), 0, S(stream), total, c, m,
                     n, points, idx, weight, out);
  return det6d_check_launch("det6d_three_interpolate");
}

DET6D_API int det6d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                           const int *idx, const float *weight, float *grad_points,
                                           det6d_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || m < 0 || !grad_out || !idx || !weight || !grad_points) return DET6D_EINVAL;
  const int64_t total = (int64_t)b * c * n;
  if (total == 0) return DET6D_OK;
  hipLaunchKernelGGL(three_interpolate_grad_kernel, grid_for(total), dim3(kBlock), 0, S(stream), total, c,
                     n, m, grad_out, idx, weight, grad_points);
  return det6d_check_launch("det6d_three_interpolate_grad");
}

DET6D_API int det6d_pack_points(int total, int cin, const float *points, int ld, float *rows,
                                float *xyz_out, det6d_stream_t stream) {
  if (total < 0 || cin < 0 || ld < 3 + cin || !points || !rows) return DET6D_EINVAL;
  const int64_t elems = (int64_t)total * ld;
  if (elems == 0) return DET6D_OK;
  hipLaunchKernelGGL(pack_points_kernel, grid_for(elems), dim3(kBlock), 0, S(stream), elems, 3 + cin, ld,
                     points, rows, xyz_out);
  return det6d_check_launch("det6d_pack_points");
}

DET6D_API int det6d_gather_centres(int b, int n, int m, const float *xyz, const int *idx, int idx_stride, int idx_bias,
                                   float *xyz_out, float *rows_out, int ld_rows, int zero_from, det6d_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0 || !xyz || !idx || !xyz_out || idx_stride < m) return DET6D_EINVAL;
  if (rows_out && (ld_rows < 3 || zero_from < 3 || zero_from > ld_rows)) return DET6D_EINVAL;
  const int per = 3 + (rows_out ? ld_rows - zero_from : 0);
  const int64_t total = (int64_t)b * m * per;
  if (total == 0) return DET6D_OK;
  hipLaunchKernelGGL(gather_centres_kernel, grid_for(total), dim3(kBlock), 0, S(stream), total, n, m, idx_stride, idx_bias, ld_rows,
                     zero_from, xyz, idx, xyz_out, rows_out);
  return det6d_check_launch("det6d_gather_centres");
}

DET6D_API int det6d_with_batch_index(int b, int m, const float *src, int ld_src, int ncol, float *dst,
                                     det6d_stream_t stream) {
  if (b < 0 || m < 0 || ncol <= 0 || ncol > ld_src || !src || !dst) return DET6D_EINVAL;
  const int64_t total = (int64_t)b * m * (ncol + 1);
  if (total == 0) return DET6D_OK;
  hipLaunchKernelGGL(with_batch_index_kernel, grid_for(total), dim3(kBlock), 0, S(stream), total, m, ld_src, ncol,
                     src, dst);
  return det6d_check_launch("det6d_with_batch_index");
}

DET6D_API int det6d_gather_rows(int b, int n, int m, int ld_in, int ld_out, int ncol, const float *rows_in,
                                const int *idx, float *rows_out, det6d_stream_t stream) {
  if (b < 0 || n < 0 || m < 0 || ncol < 0 || ncol > ld_in || ncol > ld_out || !rows_in || !idx || !rows_out)
    return DET6D_EINVAL;
  const int64_t total = (int64_t)b * m * ncol;
  if (total == 0) return DET6D_OK;
  hipLaunchKernelGGL(gather_rows_kernel, grid_for(total), dim3(kBlock), 0, S(stream), total, n, m, ld_in,
                     ld_out, ncol, rows_in, idx, rows_out);
  return det6d_check_launch("det6d_gather_rows");
}

DET6D_API int det6d_group_maxpool(int groups, int ns, int ncols, const float *x, int ldx, const int *cnt, float *y, int ldy,
                                  int col0, det6d_stream_t stream) {
  if (groups < 0 || ns <= 0 || ncols <= 0 || ncols > ldx || col0 < 0 || col0 + ncols > ldy || !x || !y) return DET6D_EINVAL;
  const int64_t total = (int64_t)groups * ncols;
  if (total == 0) return DET6D_OK;
  hipLaunchKernelGGL(group_maxpool_kernel, grid_for(total), dim3(kBlock), 0, S(stream), total, ns, ncols, x, ldx, cnt, y, ldy, col0);
  return det6d_check_launch("det6d_group_maxpool");
}

DET6D_API int det6d_sigmoid_pow(int count, const float *scores, float gamma, float *weights,
                                det6d_stream_t stream) {
  if (count < 0 || !scores || !weights) return DET6D_EINVAL;
  if (count == 0) return DET6D_OK;
  hipLaunchKernelGGL(sigmoid_pow_kernel, grid_for(count), dim3(kBlock), 0, S(stream), (int64_t)count,
                     scores, gamma, weights);
  return det6d_check_launch("det6d_sigmoid_pow");
}

DET6D_API int det6d_vote_points(int rows, const float *off, int ldo, const float *cand, int ldc, float rx,
                                float ry, float rz, float *vote, int ldv, float *off_out,
                                det6d_stream_t stream) {
  if (rows < 0 || !off || !cand || !vote || ldo < 3 || ldc < 3 || ldv < 3) return DET6D_EINVAL;
  if (rows == 0) return DET6D_OK;
  const int64_t total = (int64_t)rows * 3;
  hipLaunchKernelGGL(vote_points_kernel, grid_for(total), dim3(kBlock), 0, S(stream), total, off, ldo,
                     cand, ldc, rx, ry, rz, vote, ldv, off_out);
  return det6d_check_launch("det6d_vote_points");
}

DET6D_API int det6d_decode_boxes(int rows, int nbin, int ground_aware, int minus, float threshold_rad,
                                 float factor_rad, const float *code, int ldcode, const float *pts, int ldp,
                                 float *boxes, det6d_stream_t stream) {
  if (rows < 0 || nbin <= 0 || !code || !pts || !boxes || ldp < 3 ||
      ldcode < 6 + 2 * nbin + (ground_aware ? 2 : 1))
    return DET6D_EINVAL;
  if (rows == 0) return DET6D_OK;
  const float per_bin = (float)(3.14159265358979323846 * 2.0 / (double)nbin);
  hipLaunchKernelGGL(decode_boxes_kernel, grid_for(rows), dim3(kBlock), 0, S(stream), rows, nbin,
                     ground_aware, minus, threshold_rad, factor_rad, per_bin, code, ldcode, pts, ldp, boxes);
  return det6d_check_launch("det6d_decode_boxes");
}
