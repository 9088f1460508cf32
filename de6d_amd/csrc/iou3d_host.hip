// iou3d_host.hip — the one HOST entry of the reference's iou3d_nms interface: boxes_iou_bev_cpu
// (core/pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:16, iou3d_cpu.cpp:232-252; caller boxes_bev_iou_cpu,
// iou3d_nms_utils.py:12-29).  The reference itself runs this one on the CPU over host tensors, so it is part of the
// drop-in surface, not a fallback of a device op: host pointers in, host pointers out, no stream.
// Same geometry header as the device kernels (include/det6d_geom.h compiles for both sides).
#include "common.h"
#include "../../include/det6d_geom.h"

DET6D_API int det6d_boxes_iou_bev_cpu(int num_a, const float *boxes_a_host, int num_b, const float *boxes_b_host,
                                      float *ans_iou_host) {
  if (num_a < 0 || num_b < 0) return DET6D_EINVAL;
  if (num_a == 0 || num_b == 0) return DET6D_OK;
  if (!boxes_a_host || !boxes_b_host || !ans_iou_host) return DET6D_EINVAL;
  for (int i = 0; i < num_a; ++i)
    for (int j = 0; j < num_b; ++j)
      ans_iou_host[(size_t)i * num_b + j] = d6_iou_bev(boxes_a_host + (size_t)i * 7, boxes_b_host + (size_t)j * 7);
  return DET6D_OK;
}
