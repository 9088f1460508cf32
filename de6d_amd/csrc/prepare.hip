// prepare.hip — the input producer of the hot path on the device.
//
// One launch turns B raw LiDAR frames (concatenated (total_raw, C) rows + per-scene offsets) into the
// model's `points (B*N, 1+C)` tensor [b, x, y, z, feat..]:
//   mask_points_and_boxes_outside_range  core/pcdet/datasets/processor/data_processor.py:78-90
//                                        (common_utils.mask_points_by_range: x and y only, utils/common_utils.py:61-64)
//   sample_points                        data_processor.py:145-178 (near / far 40 m rule, duplicate padding, shuffle)
//   collate_batch 'points' branch        core/pcdet/datasets/dataset.py:171-176 (batch-index prefix column)
//   load_data_to_gpu                     core/pcdet/models/__init__.py:23-34 (the H2D copy is the caller's, of RAW frames)
// The reference does this per frame in NumPy inside DataLoader workers with the global Mersenne
// Twister; here the selection RULE is kept and the draws come from keyed bijections (include/det6d_rng.h),
// so the result is reproducible from (seed, scene) and needs no sort: a prefix count gives every point
// its rank among the in-range / near points, perm(rank) < k decides membership and IS the pre-shuffle
// slot, and a second bijection over [0, N) is the shuffle.
//
// Each scene is cut into kSeg segments, one 256-thread workgroup per segment: `prep_count_kernel`
// counts in-range / near points per segment, `prep_scatter_kernel` turns the segment counts into its
// rank base, ranks its points (wave ballots + a 4-entry LDS scan per 256-point chunk) and scatters the
// rows, `prep_pad_kernel` adds the with-replacement duplicates of very sparse frames.  HBM-bound by
// construction: 2 reads of the raw frame + one write of N rows.
#include "common.h"
#include "../../include/det6d_rng.h"

namespace {

struct PrepArgs {
  int c, num_points;
  const int *offsets;
  const int *scene_ids;  // key of each scene's random streams (NULL: position in the batch)
  const float *raw;
  float x0, y0, x1, y1, near_depth;
  uint64_t seed;
  int *rank_to_raw;  // workspace (total_raw): compacted in-range list per scene (with-replacement padding)
  int *seg_counts;   // workspace (b, kSeg, 2): in-range / near points per segment
  float *out;
  int *n_in_range;
};

__device__ __forceinline__ void classify(const float *p, const PrepArgs &a, bool &in, bool &near) {
  const float x = p[0], y = p[1], z = p[2];
  in = (x >= a.x0) & (x <= a.x1) & (y >= a.y0) & (y <= a.y1);
  // np.linalg.norm(points[:, 0:3], axis=1) in float32: sqrt((x*x + y*y) + z*z), correctly rounded sqrt
  const float d = __fsqrt_rn((x * x + y * y) + z * z);
  near = d < a.near_depth;
}

__device__ __forceinline__ void put_row(const PrepArgs &a, int scene, uint32_t slot, uint32_t key_shuffle,
                                        const float *src) {
  const uint32_t pos = d6_perm(slot, (uint32_t)a.num_points, key_shuffle);
  float *dst = a.out + ((size_t)scene * a.num_points + pos) * (1 + a.c);
  dst[0] = (float)scene;
  for (int j = 0; j < a.c; ++j) dst[1 + j] = src[j];
}

constexpr int kSeg = 32;       // segments (workgroups) per scene
constexpr int kThreads = 256;

// points [seg_lo, seg_hi) of the scene belong to this workgroup; multiples of kThreads keep chunks aligned
__device__ __forceinline__ void segment_bounds(int n_raw, int seg, int &seg_lo, int &seg_hi) {
  const int len = ((n_raw + kSeg - 1) / kSeg + kThreads - 1) / kThreads * kThreads;
  seg_lo = min(seg * len, n_raw);
  seg_hi = min(seg_lo + len, n_raw);
}

__global__ __launch_bounds__(kThreads) void prep_count_kernel(const PrepArgs a) {
  __shared__ int s_cnt[2][kThreads / 64];
  const int scene = blockIdx.y, seg = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = a.offsets[scene], n_raw = a.offsets[scene + 1] - lo;
  int seg_lo, seg_hi;
  segment_bounds(n_raw, seg, seg_lo, seg_hi);
  int c_in = 0, c_near = 0;
  for (int i = seg_lo + tid; i < seg_hi; i += kThreads) {
    bool in, near;
    classify(a.raw + (size_t)(lo + i) * a.c, a, in, near);
    c_in += in;
    c_near += in & near;
  }
  for (int off = 32; off; off >>= 1) {
    c_in += __shfl_xor(c_in, off);
    c_near += __shfl_xor(c_near, off);
  }
  if (lane == 0) { s_cnt[0][wave] = c_in; s_cnt[1][wave] = c_near; }
  __syncthreads();
  if (tid == 0) {
    int t0 = 0, t1 = 0;
    for (int w = 0; w < kThreads / 64; ++w) { t0 += s_cnt[0][w]; t1 += s_cnt[1][w]; }
    a.seg_counts[((size_t)scene * kSeg + seg) * 2 + 0] = t0;
    a.seg_counts[((size_t)scene * kSeg + seg) * 2 + 1] = t1;
  }
}

// totals of the scene and the rank base of segment `seg` (every thread computes them: 2*kSeg cached loads)
__device__ __forceinline__ void scene_totals(const PrepArgs &a, int scene, int seg, int &n_in, int &n_near,
                                             int &base_in, int &base_near) {
  n_in = n_near = base_in = base_near = 0;
  for (int s = 0; s < kSeg; ++s) {
    const int ci = a.seg_counts[((size_t)scene * kSeg + s) * 2 + 0];
    const int cn = a.seg_counts[((size_t)scene * kSeg + s) * 2 + 1];
    if (s < seg) { base_in += ci; base_near += cn; }
    n_in += ci; n_near += cn;
  }
}

__global__ __launch_bounds__(kThreads) void prep_scatter_kernel(const PrepArgs a) {
  __shared__ int s_cnt[2][kThreads / 64];
  const int scene = blockIdx.y, seg = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = a.offsets[scene], n_raw = a.offsets[scene + 1] - lo;
  const int N = a.num_points;
  int n_in, n_near, base_in, base_near;
  scene_totals(a, scene, seg, n_in, n_near, base_in, base_near);
  const int n_far = n_in - n_near;
  if (seg == 0 && tid == 0) a.n_in_range[scene] = n_in;

  if (n_in == 0) {  // nothing to sample from (the reference would raise): zero rows, batch index set
    for (int s = seg * kThreads + tid; s < N; s += kSeg * kThreads) {
      float *dst = a.out + ((size_t)scene * N + s) * (1 + a.c);
      dst[0] = (float)scene;
      for (int j = 0; j < a.c; ++j) dst[1 + j] = 0.f;
    }
    return;
  }

  const uint32_t sid = a.scene_ids ? (uint32_t)a.scene_ids[scene] : (uint32_t)scene;
  const uint32_t key_sel = d6_stream_key(a.seed, sid, 1);
  const uint32_t key_shuffle = d6_stream_key(a.seed, sid, 2);
  // the reference's three branches (data_processor.py:152-176)
  const bool subsample = N < n_in;
  const bool keep_far = subsample && N > n_far;        // all far points + a random subset of the near ones
  const int k_near = keep_far ? N - n_far : 0;
  const int extra = subsample ? 0 : N - n_in;          // duplicate padding
  const bool extra_with_replacement = extra > n_in;    // -> prep_pad_kernel

  int seg_lo, seg_hi;
  segment_bounds(n_raw, seg, seg_lo, seg_hi);
  for (int c0 = seg_lo; c0 < seg_hi; c0 += kThreads) {
    const int i = c0 + tid;
    bool in = false, near = false;
    const float *src = a.raw + (size_t)(lo + (i < seg_hi ? i : seg_lo)) * a.c;
    if (i < seg_hi) classify(src, a, in, near);
    near = near & in;
    const uint64_t b_in = __ballot(in), b_near = __ballot(near);
    const uint64_t lt = ((uint64_t)1 << lane) - 1;
    if (lane == 0) { s_cnt[0][wave] = __popcll(b_in); s_cnt[1][wave] = __popcll(b_near); }
    __syncthreads();
    int w_in = 0, w_near = 0, t_in = 0, t_near = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) {
      const int ci = s_cnt[0][w], cn = s_cnt[1][w];
      if (w < wave) { w_in += ci; w_near += cn; }
      t_in += ci; t_near += cn;
    }
    if (in) {
      const int r_in = base_in + w_in + __popcll(b_in & lt);
      const int r_near = base_near + w_near + __popcll(b_near & lt);
      const int r_far = r_in - r_near;
      a.rank_to_raw[lo + r_in] = i;
      if (subsample) {
        if (keep_far) {
          if (near) {
            const uint32_t p = d6_perm((uint32_t)r_near, (uint32_t)n_near, key_sel);
            if (p < (uint32_t)k_near) put_row(a, scene, p, key_shuffle, src);
          } else {
            put_row(a, scene, (uint32_t)(k_near + r_far), key_shuffle, src);
          }
        } else {
          const uint32_t p = d6_perm((uint32_t)r_in, (uint32_t)n_in, key_sel);
          if (p < (uint32_t)N) put_row(a, scene, p, key_shuffle, src);
        }
      } else {
        put_row(a, scene, (uint32_t)r_in, key_shuffle, src);
        if (extra > 0 && !extra_with_replacement) {
          const uint32_t p = d6_perm((uint32_t)r_in, (uint32_t)n_in, key_sel);
          if (p < (uint32_t)extra) put_row(a, scene, (uint32_t)n_in + p, key_shuffle, src);
        }
      }
    }
    base_in += t_in;
    base_near += t_near;
    __syncthreads();
  }
}

// padding drawn with replacement (fewer than N/2 points in range): needs the complete rank -> raw
// index list of the scene, hence its own launch after prep_scatter_kernel
__global__ __launch_bounds__(kThreads) void prep_pad_kernel(const PrepArgs a) {
  const int scene = blockIdx.y, seg = blockIdx.x, tid = threadIdx.x;
  const int lo = a.offsets[scene];
  const int N = a.num_points;
  int n_in, n_near, base_in, base_near;
  scene_totals(a, scene, 0, n_in, n_near, base_in, base_near);
  const int extra = N - n_in;
  if (n_in == 0 || extra <= n_in) return;
  const uint32_t sid = a.scene_ids ? (uint32_t)a.scene_ids[scene] : (uint32_t)scene;
  const uint32_t key_shuffle = d6_stream_key(a.seed, sid, 2);
  const uint32_t key_extra = d6_stream_key(a.seed, sid, 3);
  for (int e = seg * kThreads + tid; e < extra; e += kSeg * kThreads) {
    const uint32_t q = d6_randint((uint32_t)e, (uint32_t)n_in, key_extra);
    const int i = a.rank_to_raw[lo + q];
    put_row(a, scene, (uint32_t)(n_in + e), key_shuffle, a.raw + (size_t)(lo + i) * a.c);
  }
}

}  // namespace

DET6D_API int64_t det6d_prepare_points_workspace_bytes(int b, int total_raw) {
  return ((int64_t)(total_raw > 0 ? total_raw : 1) + (int64_t)(b > 0 ? b : 1) * kSeg * 2) * (int64_t)sizeof(int);
}

DET6D_API int det6d_prepare_points(int b, const int *raw_offsets, const int *scene_ids, int total_raw, int c,
                                   const float *raw,
                                   float x_min, float y_min, float x_max, float y_max, int num_points,
                                   float near_depth, uint64_t seed, void *workspace, float *points_out,
                                   int *n_in_range, det6d_stream_t stream) {
  if (b < 0 || total_raw < 0 || c < 3 || num_points <= 0) return DET6D_EINVAL;
  if (b == 0) return DET6D_OK;
  if (!raw_offsets || (!raw && total_raw > 0) || !workspace || !points_out || !n_in_range) return DET6D_EINVAL;
  PrepArgs a;
  a.c = c; a.num_points = num_points; a.offsets = raw_offsets; a.scene_ids = scene_ids; a.raw = raw;
  a.x0 = x_min; a.y0 = y_min; a.x1 = x_max; a.y1 = y_max; a.near_depth = near_depth;
  a.seed = seed; a.rank_to_raw = (int *)workspace; a.seg_counts = a.rank_to_raw + (total_raw > 0 ? total_raw : 1);
  a.out = points_out; a.n_in_range = n_in_range;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(prep_count_kernel, dim3(kSeg, b), dim3(kThreads), 0, s, a);
  hipLaunchKernelGGL(prep_scatter_kernel, dim3(kSeg, b), dim3(kThreads), 0, s, a);
  hipLaunchKernelGGL(prep_pad_kernel, dim3(kSeg, b), dim3(kThreads), 0, s, a);
  return det6d_check_launch("det6d_prepare_points");
}
