// iou3d_nms.hip — rotated BEV IoU and NMS for gfx950.
//
// Replaces core/pcdet/ops/iou3d_nms/src/{iou3d_nms_kernel.cu:236-372, iou3d_nms.cpp:49-186}.
//  * suppression bit-matrix: one wave64 per 64x64 tile, lane j evaluates iou(box_i, box_j) for a
//    wave-uniform row i and __ballot() yields row i's 64-bit word directly — the same
//    `unsigned long long` layout the reference builds bit by bit (iou3d_nms_kernel.cu:298-309);
//  * greedy scan: one wave on the device (lane w owns removal word w), so there is no
//    cudaMalloc / D2H copy / host loop / cudaFree per scene as in iou3d_nms.cpp:102-132;
//  * det6d_postprocess fuses score filter + stable sort + NMS + selection per scene in one
//    workgroup (detector3d_template.py:178-284, model_nms_utils.py:6-25, iou3d_nms_utils.py:84-99).
#include "common.h"
#include "../../include/det6d_geom.h"

#include <vector>

namespace {

// ---- separated pairs ------------------------------------------------------------------------------------------------------
// A suppression bit is `iou_bev(a, b) > thresh` (iou3d_nms_kernel.cu:267-311).  The rotated overlap behind it is ~3 000 vector
// instructions per pair (16 edge tests, 8 corner tests, a bubble sort by atan2), and almost all pairs of a scene's candidates are
// far apart.  Two boxes whose centres are farther apart than the sum of their CIRCUMRADII (plus a margin) cannot produce a set
// bit in the reference's arithmetic:
//   * no corner of one passes check_in_box2d of the other (iou3d_cpu.cpp:75-85: its 1e-2 margin grows a box by at most
//     1.5e-2 in any direction; the radius below carries 5e-2 + 1e-4 relative + 1e-5 of the coordinates for that and rounding);
//   * a segment pair contributes a point only if s1 * s2 > 0 && s3 * s4 > 0 (iou3d_cpu.cpp:87-103); for disjoint convex
//     boxes that takes a rounding-flipped sign of a cross product of (near-)collinear edges, and a non-zero area needs THREE
//     such points with finite coordinates — the collinear branch divides by D = a0 * b1 - a1 * b0 ~ 0, so what such pairs
//     yield is inf / NaN corner sums, a NaN area and `NaN > thresh` == false, which is what skipping the pair gives too.
// tests/test_nms_filter.py runs the reference's own iou3d_cpu.cpp over millions of separated pairs, rows of collinear
// axis-aligned boxes included: no pair the rule skips has an IoU above 0 there.  The rule is used where only the BIT matters (mask
// kernels), never in det6d_boxes_iou_bev / _overlap_bev, whose matrices are the reference's value for every pair.  thresh < 0
// (nothing real) switches it off.
__device__ __forceinline__ float d6_nms_radius(float x, float y, float dx, float dy) {
  return 0.5f * sqrtf(dx * dx + dy * dy) * 1.0001f + 0.05f + 1e-5f * (fabsf(x) + fabsf(y));
}
// false only when the pair is provably separated (any NaN / inf leaves it true: the pair is evaluated in full)
__device__ __forceinline__ bool d6_nms_near(float xa, float ya, float ra, float xb, float yb, float rb) {
  const float ex = xa - xb, ey = ya - yb, s = ra + rb;
  return !(ex * ex + ey * ey > s * s);
}

template <bool IOU>
__global__ void boxes_pair_kernel(int num_a, const float *__restrict__ boxes_a, int num_b,
                                  const float *__restrict__ boxes_b, float *__restrict__ ans) {
  const int a_idx = blockIdx.y * 16 + threadIdx.y;
  const int b_idx = blockIdx.x * 16 + threadIdx.x;
  if (a_idx >= num_a || b_idx >= num_b) return;
  float ba[7], bb[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) { ba[c] = boxes_a[a_idx * 7 + c]; bb[c] = boxes_b[b_idx * 7 + c]; }
  ans[(size_t)a_idx * num_b + b_idx] = IOU ? d6_iou_bev(ba, bb) : d6_box_overlap(ba, bb);
}

// One wave per (box i, 64-column block c): lane j evaluates iou(box_i, box_{64c+j}); the wave ballot IS the reference's
// 64-bit mask word (iou3d_nms_kernel.cu:267-311 walks the 64 rows of a 64x64 tile in one thread block: on a chip with
// 1024 SIMDs that serialises 64 rotated-IoU evaluations per wave and leaves most of it idle at K = 256: 1.1 ms).  The grid
// is (K, column blocks) waves, four per workgroup; blocks below the diagonal are never read by the greedy scan.
template <bool NORMAL>
__global__ __launch_bounds__(256) void nms_mask_kernel(int boxes_num, float thresh,
                                                       const float *__restrict__ boxes,
                                                       unsigned long long *__restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int col_start = blockIdx.y;
  const int col_blocks = (boxes_num + 63) / 64;
  if (i >= boxes_num || col_start < (i >> 6)) return;      // wave-uniform
  const int j = col_start * 64 + lane;
  float bi[7], bj[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) bi[c] = boxes[(size_t)i * 7 + c];  // uniform address -> scalar loads
#pragma unroll
  for (int c = 0; c < 7; ++c) bj[c] = j < boxes_num ? boxes[(size_t)j * 7 + c] : 0.f;
  bool sup = false;
  // (separated pairs cannot set a bit: see d6_nms_near; the axis-aligned iou_normal is cheap and evaluated for every pair)
  const bool live = j > i && j < boxes_num &&
                    (NORMAL || thresh < 0.f ||
                     d6_nms_near(bi[0], bi[1], d6_nms_radius(bi[0], bi[1], bi[3], bi[4]), bj[0], bj[1], d6_nms_radius(bj[0], bj[1], bj[3], bj[4])));
  if (live) {
    const float v = NORMAL ? d6_iou_normal(bi, bj) : d6_iou_bev(bi, bj);
    sup = v > thresh;
  }
  const unsigned long long word = __ballot(sup);
  if (lane == 0) mask[(size_t)i * col_blocks + col_start] = word;
}

// Greedy scan, iou3d_nms.cpp:116-132.  Lane w owns remv[w]; boxes beyond 64*64 loop over words.
__global__ __launch_bounds__(64) void nms_greedy_kernel(int boxes_num, int chunk_rows,
                                                        const unsigned long long *__restrict__ mask,
                                                        long long *__restrict__ keep,
                                                        int *__restrict__ num_keep) {
  // the greedy scan is a chain of dependent reads of suppression rows: the rows are staged into LDS `chunk_rows` at a
  // time with plain coalesced loads (all in flight at once), so a step costs an LDS read instead of a ~2 us round trip
  // to memory (256 boxes: 1.2 ms -> ~0.05 ms)
  extern __shared__ unsigned long long lds_nms[];
  const int lane = threadIdx.x;
  const int col_blocks = (boxes_num + 63) / 64;
  unsigned long long *remv_s = lds_nms, *rows_s = lds_nms + col_blocks;
  for (int w = lane; w < col_blocks; w += 64) remv_s[w] = 0ull;
  int kept = 0;
  for (int r0 = 0; r0 < boxes_num; r0 += chunk_rows) {
    const int nrow = boxes_num - r0 < chunk_rows ? boxes_num - r0 : chunk_rows;
    __syncthreads();
    for (int e = lane; e < nrow * col_blocks; e += 64) rows_s[e] = mask[(size_t)r0 * col_blocks + e];
    __syncthreads();
    for (int i = r0; i < r0 + nrow; ++i) {
      const int nblock = i >> 6, inblock = i & 63;
      const unsigned long long cur = remv_s[nblock];  // uniform
      if (!((cur >> inblock) & 1ull)) {
        if (lane == 0) keep[kept] = i;
        ++kept;
        for (int w = nblock + lane; w < col_blocks; w += 64) remv_s[w] |= rows_s[(size_t)(i - r0) * col_blocks + w];
        __syncthreads();
      }
    }
  }
  if (lane == 0) *num_keep = kept;
}

template <bool NORMAL>
int run_nms(int boxes_num, const float *boxes, float thresh, uint64_t *mask, int64_t *keep,
            int *num_keep, hipStream_t stream) {
  if (boxes_num < 0 || !num_keep || (boxes_num > 0 && (!boxes || !mask || !keep))) return DET6D_EINVAL;
  const int col_blocks = (boxes_num + 63) / 64;
  if (boxes_num == 0) {
    hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int), stream);
    if (e != hipSuccess) { det6d_set_error("det6d_nms memset", e); return DET6D_ELAUNCH; }
    return DET6D_OK;
  }
  hipLaunchKernelGGL((nms_mask_kernel<NORMAL>), dim3((boxes_num + 3) / 4, col_blocks), dim3(256), 0, stream,
                     boxes_num, thresh, boxes, (unsigned long long *)mask);
  // rows staged per chunk: as many as fit 96 KB of LDS beside the removal words
  int chunk_rows = (int)((96 * 1024) / ((size_t)col_blocks * sizeof(unsigned long long)));
  if (chunk_rows > boxes_num) chunk_rows = boxes_num;
  if (chunk_rows < 1) chunk_rows = 1;
  const size_t greedy_lds = ((size_t)col_blocks + (size_t)chunk_rows * col_blocks) * sizeof(unsigned long long);
  if (greedy_lds > 128 * 1024) return DET6D_EINVAL;   // > ~390 k boxes
  static bool greedy_attr = false;
  if (!greedy_attr) {
    hipFuncSetAttribute((const void *)nms_greedy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    greedy_attr = true;
  }
  hipLaunchKernelGGL(nms_greedy_kernel, dim3(1), dim3(64), greedy_lds, stream, boxes_num, chunk_rows,
                     (const unsigned long long *)mask, (long long *)keep, num_keep);
  return det6d_check_launch("det6d_nms");
}

// ------------------------------------------------------------------------------------------
// Per-scene post-processing in three short launches over a caller-provided workspace
// (P <= 1024 candidates per scene):
//   post_rank_kernel   one workgroup per scene: sigmoid, class max, score filter, stable rank
//                      sort by counting, gather of the sorted boxes;
//   post_mask_kernel   one WAVE per (sorted row, 64-column block) of the upper-triangular
//                      suppression matrix, spread over the whole chip (the 64 x 64 tile loop of the
//                      stand-alone NMS serialises 64 heavy IoU evaluations per wave);
//   post_select_kernel one wave per scene: greedy scan over the mask (staged in LDS) + outputs.
// ------------------------------------------------------------------------------------------
constexpr int kPostThreads = 256;
constexpr int kPostMaxP = 1024;
constexpr int kPostCB = kPostMaxP / 64;

struct PostWs {          // layout of the workspace, per scene
  int cand;
  int pad[15];
  float score[kPostMaxP];
  int label[kPostMaxP];
  int order[kPostMaxP];
  float sorted[kPostMaxP * 8];                 // first 7 = box dims used by NMS
  unsigned long long mask[kPostMaxP * kPostCB];
  float4 key[kPostMaxP];                       // (x, y, circumradius + margin, -) of the sorted boxes: the separated-pair test
};

__global__ __launch_bounds__(kPostThreads) void post_rank_kernel(int p, int ncls, const float *__restrict__ cls,
                                                                 const float *__restrict__ boxes, float score_thr,
                                                                 int pre_max, PostWs *__restrict__ ws_all) {
  __shared__ float s_score[kPostMaxP];
  __shared__ int s_cand;
  const int bi = blockIdx.x, tid = threadIdx.x;
  PostWs &ws = ws_all[bi];
  cls += (size_t)bi * p * ncls;
  boxes += (size_t)bi * p * 9;
  if (tid == 0) s_cand = 0;
  for (int i = tid; i < p; i += kPostThreads) {
    float best = d6_sigmoidf(cls[(size_t)i * ncls]);
    int bl = 0;
    for (int c = 1; c < ncls; ++c) {
      const float s = d6_sigmoidf(cls[(size_t)i * ncls + c]);
      if (s > best) { best = s; bl = c; }
    }
    s_score[i] = best;
    ws.score[i] = best;
    ws.label[i] = bl + 1;
  }
  __syncthreads();
  for (int i = tid; i < p; i += kPostThreads) {
    const float si = s_score[i];
    if (si >= score_thr) {
      int rank = 0;
      for (int j = 0; j < p; ++j) {
        const float sj = s_score[j];
        rank += (sj >= score_thr) && (sj > si || (sj == si && j < i));
      }
      ws.order[rank] = i;
#pragma unroll
      for (int c = 0; c < 7; ++c) ws.sorted[rank * 8 + c] = boxes[(size_t)i * 9 + c];
      const float bx = boxes[(size_t)i * 9 + 0], by = boxes[(size_t)i * 9 + 1];
      ws.key[rank] = make_float4(bx, by, d6_nms_radius(bx, by, boxes[(size_t)i * 9 + 3], boxes[(size_t)i * 9 + 4]), 0.f);
      atomicAdd(&s_cand, 1);
    }
  }
  __syncthreads();
  if (tid == 0) ws.cand = min(s_cand, pre_max);
}

// Suppression rows of kMaskRows consecutive sorted rows of one scene per workgroup, in two phases (round 5):
//   A  every (row, column) pair of the chunk above the diagonal through the separated-pair test (6 vector instructions per 64
//      pairs); the pairs that survive are appended to a queue in LDS (16-bit entries: local row | column);
//   B  the queue is evaluated DENSELY — one rotated IoU per lane with all lanes live — and the set bits are OR-ed into the rows
//      in LDS.
// Round 4 evaluated every pair in place: on 65536-point scenes (1024 candidates, 524 k pairs per scene, 76 % of the pipeline's
// workgroup-time) a wave ran the 3 000-instruction overlap for 64 pairs of which a handful were near each other.
constexpr int kMaskRows = 8;

__global__ __launch_bounds__(256) void post_mask_kernel(float nms_thr, PostWs *__restrict__ ws_all) {
  __shared__ unsigned short queue[kMaskRows * kPostMaxP];
  __shared__ unsigned int rows_s[kMaskRows][2 * kPostCB];
  __shared__ float4 rkey[kMaskRows];
  __shared__ int q_count;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  PostWs &ws = ws_all[blockIdx.y];
  const int cand = ws.cand;
  const int r0 = blockIdx.x * kMaskRows;
  if (r0 >= cand) return;
  const int nrow = min(kMaskRows, cand - r0);
  const int cb0 = r0 >> 6, cb1 = (cand + 63) >> 6;                 // live column blocks of this chunk
  if (tid == 0) q_count = 0;
  if (tid < kMaskRows) rkey[tid] = ws.key[min(r0 + tid, cand - 1)];
  for (int e = tid; e < kMaskRows * 2 * kPostCB; e += 256) (&rows_s[0][0])[e] = 0u;
  __syncthreads();
  const bool filter = nms_thr >= 0.f;
  // ---- A
  for (int c = cb0 + wave; c < cb1; c += 4) {
    const int j = c * 64 + lane;
    const float4 kj = ws.key[min(j, cand - 1)];
    for (int r = 0; r < nrow; ++r) {
      const float4 ki = rkey[r];
      const bool near = j > r0 + r && j < cand && (!filter || d6_nms_near(ki.x, ki.y, ki.z, kj.x, kj.y, kj.z));
      const unsigned long long m = __ballot(near);
      if (m != 0ull) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&q_count, __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (near) queue[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)((r << 10) | j);
      }
    }
  }
  __syncthreads();
  // ---- B
  const int total = q_count;
  for (int q = tid; q < total; q += 256) {
    const int e = queue[q], r = e >> 10, j = e & 1023;
    float bi[7], bj[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) { bi[c] = ws.sorted[(r0 + r) * 8 + c]; bj[c] = ws.sorted[j * 8 + c]; }
    if (d6_iou_bev(bi, bj) > nms_thr) atomicOr(&rows_s[r][j >> 5], 1u << (j & 31));
  }
  __syncthreads();
  for (int e = tid; e < nrow * (cb1 - cb0); e += 256) {
    const int r = e / (cb1 - cb0), c = cb0 + e % (cb1 - cb0);
    ws.mask[(r0 + r) * kPostCB + c] = (unsigned long long)rows_s[r][2 * c] | ((unsigned long long)rows_s[r][2 * c + 1] << 32);
  }
}

__global__ __launch_bounds__(64) void post_select_kernel(int p, const float *__restrict__ boxes, int post_max,
                                                         const PostWs *__restrict__ ws_all,
                                                         float *__restrict__ out_boxes, float *__restrict__ out_scores,
                                                         int *__restrict__ out_labels, int *__restrict__ out_index,
                                                         int *__restrict__ out_count) {
  extern __shared__ unsigned long long s_mask[];   // p * ceil(p/64) words (dynamic: 8 KB at p=256, 128 KB at 1024)
  __shared__ unsigned long long s_remv[kPostCB];
  __shared__ int s_keep[kPostMaxP];
  const int bi = blockIdx.x, lane = threadIdx.x;
  const PostWs &ws = ws_all[bi];
  boxes += (size_t)bi * p * 9;
  const int cand = ws.cand;
  const int col_blocks = (cand + 63) / 64;
  const int ldm = (p + 63) / 64;                   // LDS row stride in words
  // stage the rows' live words (own block onward) in LDS: the scan below is a serial chain
  for (int t = lane; t < cand * ldm; t += 64) {
    const int i = t / ldm, w = t % ldm;
    s_mask[t] = (w >= (i >> 6) && w < col_blocks) ? ws.mask[i * kPostCB + w] : 0ull;
  }
  if (lane < kPostCB) s_remv[lane] = 0ull;
  __syncthreads();
  int kept = 0;
  for (int i = 0; i < cand; ++i) {
    const int nblock = i >> 6, inblock = i & 63;
    const unsigned long long cur = s_remv[nblock];
    if (!((cur >> inblock) & 1ull)) {
      if (lane == 0) s_keep[kept] = i;
      ++kept;
      if (lane >= nblock && lane < col_blocks) s_remv[lane] |= s_mask[i * ldm + lane];
      __syncthreads();
    }
  }
  __syncthreads();
  const int nkeep = min(kept, post_max);
  if (lane == 0) out_count[bi] = nkeep;
  for (int i = lane; i < post_max; i += 64) {
    float *ob = out_boxes + ((size_t)bi * post_max + i) * 9;
    if (i < nkeep) {
      const int src = ws.order[s_keep[i]];
      for (int c = 0; c < 9; ++c) ob[c] = boxes[(size_t)src * 9 + c];
      out_scores[(size_t)bi * post_max + i] = ws.score[src];
      out_labels[(size_t)bi * post_max + i] = ws.label[src];
      out_index[(size_t)bi * post_max + i] = src;
    } else {
      for (int c = 0; c < 9; ++c) ob[c] = 0.f;
      out_scores[(size_t)bi * post_max + i] = 0.f;
      out_labels[(size_t)bi * post_max + i] = 0;
      out_index[(size_t)bi * post_max + i] = -1;
    }
  }
}

}  // namespace

DET6D_API int64_t det6d_nms_mask_words(int boxes_num) {
  return (int64_t)boxes_num * ((boxes_num + 63) / 64);
}

DET6D_API int det6d_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                                      float *ans_overlap, det6d_stream_t stream) {
  if (num_a < 0 || num_b < 0 || !ans_overlap) return DET6D_EINVAL;
  if (num_a == 0 || num_b == 0) return DET6D_OK;
  hipLaunchKernelGGL((boxes_pair_kernel<false>), dim3(det6d_divup(num_b, 16), det6d_divup(num_a, 16)),
                     dim3(16, 16), 0, (hipStream_t)stream, num_a, boxes_a, num_b, boxes_b, ans_overlap);
  return det6d_check_launch("det6d_boxes_overlap_bev");
}

DET6D_API int det6d_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                                  float *ans_iou, det6d_stream_t stream) {
  if (num_a < 0 || num_b < 0 || !ans_iou) return DET6D_EINVAL;
  if (num_a == 0 || num_b == 0) return DET6D_OK;
  hipLaunchKernelGGL((boxes_pair_kernel<true>), dim3(det6d_divup(num_b, 16), det6d_divup(num_a, 16)),
                     dim3(16, 16), 0, (hipStream_t)stream, num_a, boxes_a, num_b, boxes_b, ans_iou);
  return det6d_check_launch("det6d_boxes_iou_bev");
}

DET6D_API int det6d_nms(int boxes_num, const float *boxes, float thresh, uint64_t *mask, int64_t *keep,
                        int *num_keep, det6d_stream_t stream) {
  return run_nms<false>(boxes_num, boxes, thresh, mask, keep, num_keep, (hipStream_t)stream);
}

DET6D_API int det6d_nms_normal(int boxes_num, const float *boxes, float thresh, uint64_t *mask,
                               int64_t *keep, int *num_keep, det6d_stream_t stream) {
  return run_nms<true>(boxes_num, boxes, thresh, mask, keep, num_keep, (hipStream_t)stream);
}

DET6D_API int det6d_nms_to_host(int boxes_num, const float *boxes, float thresh, int64_t *keep_host,
                                int normal, det6d_stream_t stream) {
  if (boxes_num < 0 || (boxes_num > 0 && (!boxes || !keep_host))) return DET6D_EINVAL;
  if (boxes_num == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const size_t words = (size_t)det6d_nms_mask_words(boxes_num);
  void *ws = nullptr;
  const size_t bytes = words * 8 + (size_t)boxes_num * 8 + 8;
  hipError_t e = hipMalloc(&ws, bytes);
  if (e != hipSuccess) { det6d_set_error("det6d_nms_to_host hipMalloc", e); return DET6D_ELAUNCH; }
  uint64_t *mask = (uint64_t *)ws;
  int64_t *keep = (int64_t *)((char *)ws + words * 8);
  int *num = (int *)((char *)ws + words * 8 + (size_t)boxes_num * 8);
  int rc = normal ? run_nms<true>(boxes_num, boxes, thresh, mask, keep, num, s)
                  : run_nms<false>(boxes_num, boxes, thresh, mask, keep, num, s);
  int n_keep = 0;
  if (rc == DET6D_OK) {
    e = hipMemcpyAsync(&n_keep, num, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && n_keep > 0)
      e = hipMemcpy(keep_host, keep, (size_t)n_keep * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { det6d_set_error("det6d_nms_to_host copy", e); rc = DET6D_ELAUNCH; }
  }
  hipFree(ws);
  return rc == DET6D_OK ? n_keep : rc;
}

DET6D_API int64_t det6d_postprocess_workspace_bytes(int b) { return (int64_t)sizeof(PostWs) * (b > 0 ? b : 0); }

DET6D_API int det6d_postprocess(int b, int p, int ncls, const float *cls, const float *boxes,
                                float score_thr, int pre_max, int post_max, float nms_thr, void *workspace,
                                float *out_boxes, float *out_scores, int *out_labels, int *out_index,
                                int *out_count, det6d_stream_t stream) {
  if (b < 0 || p <= 0 || p > kPostMaxP || ncls <= 0 || pre_max <= 0 || post_max <= 0 || !cls || !boxes ||
      !workspace || ((uintptr_t)workspace & 15) || !out_boxes || !out_scores || !out_labels || !out_index || !out_count)
    return DET6D_EINVAL;
  if (b == 0) return DET6D_OK;
  hipStream_t s = (hipStream_t)stream;
  PostWs *ws = (PostWs *)workspace;
  hipLaunchKernelGGL(post_rank_kernel, dim3(b), dim3(kPostThreads), 0, s, p, ncls, cls, boxes, score_thr,
                     pre_max > kPostMaxP ? kPostMaxP : pre_max, ws);
  hipLaunchKernelGGL(post_mask_kernel, dim3(det6d_divup(p, kMaskRows), b), dim3(256), 0, s, nms_thr, ws);
  const size_t lds = (size_t)p * ((p + 63) / 64) * sizeof(unsigned long long);
  static bool big_lds_enabled = false;
  if (lds > 48 * 1024 && !big_lds_enabled) {   // one-time opt-in for > 64 KB of dynamic LDS (160 KB per CU on gfx950)
    hipError_t e = hipFuncSetAttribute((const void *)post_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kPostMaxP * kPostCB * 8);
    if (e != hipSuccess) { det6d_set_error("det6d_postprocess hipFuncSetAttribute", e); return DET6D_ELAUNCH; }
    big_lds_enabled = true;
  }
  hipLaunchKernelGGL(post_select_kernel, dim3(b), dim3(64), lds, s, p, boxes, post_max, ws, out_boxes, out_scores,
                     out_labels, out_index, out_count);
  return det6d_check_launch("det6d_postprocess");
}
