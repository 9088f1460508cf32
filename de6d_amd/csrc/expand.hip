// expand.hip — the FIRST layer of a grouped MLP without a GEMM over the grouped rows.
//
// Reference: every (centre, neighbour) row [x - cx, y - cy, z - cz, f_0 .. f_{C-1}] goes through Conv2d(3 + C -> C1)
// + BN + ReLU (pointnet2_utils.py:449-455, pointnet2_modules.py:561-568): rows x (3 + C) x C1 multiply-adds, 12.8 % of
// the network's flops.  With the oracle's chain order (features first, relative coordinates last: chain_k in
// oracle/det6d_oracle.c) the feature part of every output is a function of the POINT alone:
//     P[p][c]   = fma-chain_{k = 3 .. K-1}( row_p[k] * W[k][c] )                 one plain GEMM over the N points
//     out[r][c] = act( fma(dz, W[2][c], fma(dy, W[1][c], fma(dx, W[0][c], P[p(r)][c]))) + shift[c] )
// which is bit for bit the oracle's chain.  This kernel is the second line: gather P rows through the ball-query /
// compact row lists, three FMAs, shift, ReLU — HBM-bound (reads rows x C1 gathered, writes rows x C1).
#include "common.h"

namespace {

struct ExpandArgs {
  int rows;                    // dense: b * m * ns; compact: capacity (live count in hdr[0])
  int c1;                      // output columns (multiple of 4), written at out[r * ldo + 0 .. c1); [c1, ldo) zero-filled
  const float *p; int ldp; int pcol0;     // P (points, ldp), this group's columns start at pcol0
  const float *w; int ldw;     // the layer's folded weights: rows 0..2 are the coordinate rows
  const float *shift;
  int act;
  const float *pts; int ldpts; // point rows (x, y, z in columns 0..2)
  const float *ctr; int ldctr;
  // dense rows
  const int *idx; int n, m, ns;
  // compact rows
  const int *hdr; const int *crow_p; const int *crow_c;
  float *out; int ldo;
};

__global__ __launch_bounds__(256) void group_expand_kernel(const ExpandArgs g) {
  const int c4 = g.ldo >> 2;                       // float4 columns per output row (pad columns included)
  const int live = g.hdr ? g.hdr[0] : g.rows;
  const long long total = (long long)live * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / c4);
    const int c = 4 * (int)(i - (long long)r * c4);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    long long prow;
    int cj;
    bool real = true;
    if (g.hdr) {
      const int tag = g.crow_c[r];
      real = tag >= 0;                               // alignment rows of the compact list: zeros
      cj = tag & 0x1fffffff;
      prow = g.crow_p[r];
    } else {
      cj = r / g.ns;
      prow = (long long)(cj / g.m) * g.n + g.idx[r];
    }
    if (real && c < g.c1) {
      const float *pt = g.pts + prow * g.ldpts;
      const float *ce = g.ctr + (long long)cj * g.ldctr;
      const float dx = pt[0] - ce[0], dy = pt[1] - ce[1], dz = pt[2] - ce[2];
      const float4 pv = *reinterpret_cast<const float4 *>(g.p + prow * g.ldp + g.pcol0 + c);
      const float4 wx = *reinterpret_cast<const float4 *>(g.w + c);
      const float4 wy = *reinterpret_cast<const float4 *>(g.w + g.ldw + c);
      const float4 wz = *reinterpret_cast<const float4 *>(g.w + 2 * (long long)g.ldw + c);
      float4 sh = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g.shift) sh = *reinterpret_cast<const float4 *>(g.shift + c);
      o.x = D6_FMA(dz, wz.x, D6_FMA(dy, wy.x, D6_FMA(dx, wx.x, pv.x))) + sh.x;
      o.y = D6_FMA(dz, wz.y, D6_FMA(dy, wy.y, D6_FMA(dx, wx.y, pv.y))) + sh.y;
      o.z = D6_FMA(dz, wz.z, D6_FMA(dy, wy.z, D6_FMA(dx, wx.z, pv.z))) + sh.z;
      o.w = D6_FMA(dz, wz.w, D6_FMA(dy, wy.w, D6_FMA(dx, wx.w, pv.w))) + sh.w;
      if (g.act == 1) { o.x = d6_relu(o.x); o.y = d6_relu(o.y); o.z = d6_relu(o.z); o.w = d6_relu(o.w); }
    }
    *reinterpret_cast<float4 *>(g.out + (long long)r * g.ldo + c) = o;
  }
}

}  // namespace

DET6D_API int det6d_group_expand(int rows, int c1, const float *p, int ldp, int pcol0, const float *w, int ldw,
                                 const float *shift, int act, const float *pts, int ldpts, const float *ctr, int ldctr,
                                 const int *idx, int n, int m, int ns, const int *hdr, const int *crow_p, const int *crow_c,
                                 float *out, int ldo, det6d_stream_t stream) {
  if (rows < 0 || c1 <= 0 || (c1 & 3) || !p || !w || !pts || !ctr || !out) return DET6D_EINVAL;
  if ((ldp & 3) || (pcol0 & 3) || (ldw & 3) || (ldo & 3) || ldo < c1 || ldp < pcol0 + c1 || ldw < c1 || ldpts < 3 || ldctr < 3)
    return DET6D_EINVAL;
  if (((uintptr_t)p | (uintptr_t)w | (uintptr_t)out | (uintptr_t)shift) & 15) return DET6D_EINVAL;
  if (hdr ? (!crow_p || !crow_c) : (!idx || n <= 0 || m <= 0 || ns <= 0 || rows % (m * ns))) return DET6D_EINVAL;
  if (rows == 0) return DET6D_OK;
  ExpandArgs g;
  g.rows = rows; g.c1 = c1; g.p = p; g.ldp = ldp; g.pcol0 = pcol0; g.w = w; g.ldw = ldw; g.shift = shift; g.act = act;
  g.pts = pts; g.ldpts = ldpts; g.ctr = ctr; g.ldctr = ldctr; g.idx = idx; g.n = n; g.m = m; g.ns = ns;
  g.hdr = hdr; g.crow_p = crow_p; g.crow_c = crow_c; g.out = out; g.ldo = ldo;
  const long long work = (long long)rows * (ldo >> 2);
  long long blocks = (work + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;          // grid-stride: 16 workgroups per CU
  hipLaunchKernelGGL(group_expand_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
  return det6d_check_launch("det6d_group_expand");
}
