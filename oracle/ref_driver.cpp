// ref_driver.cpp — thin C entry point around the REFERENCE's own boxes_iou_bev_cpu
// (core/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232-252), which is compiled from
// /root/reference where it lies (see oracle/Makefile, target _ref).  Test infrastructure only.
#include <torch/torch.h>
#include <cstdint>

int boxes_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);

extern "C" __attribute__((visibility("default")))
int ref_boxes_iou_bev_cpu(const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* ans_iou) {
  auto opt = torch::TensorOptions().dtype(torch::kFloat32);
  at::Tensor a = torch::from_blob(const_cast<float*>(boxes_a), {num_a, 7}, opt);
  at::Tensor b = torch::from_blob(const_cast<float*>(boxes_b), {num_b, 7}, opt);
  at::Tensor o = torch::from_blob(ans_iou, {num_a, num_b}, opt);
  return boxes_iou_bev_cpu(a, b, o);
}
