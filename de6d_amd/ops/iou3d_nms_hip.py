"""Replacement for the reference extension module ``iou3d_nms_cuda``
(core/pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17; iou3d_nms.cpp:49-186).

``nms_gpu(boxes, keep, thr)`` keeps the reference contract: `boxes` (K,7) device tensor sorted by
score, `keep` a CPU int64 tensor of length K that receives the kept indices, return value =
number kept.  Internally the mask and the greedy scan both run on the device
(csrc/iou3d_nms.hip); only the short keep list crosses to the host.
"""
import ctypes

from .. import _lib as L


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    L.require_cuda(boxes_a, boxes_b, ans_overlap)
    L.call("det6d_boxes_overlap_bev", boxes_a.shape[0], L.ptr(boxes_a), boxes_b.shape[0], L.ptr(boxes_b),
           L.ptr(ans_overlap), L.stream_ptr())
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    L.require_cuda(boxes_a, boxes_b, ans_iou)
    L.call("det6d_boxes_iou_bev", boxes_a.shape[0], L.ptr(boxes_a), boxes_b.shape[0], L.ptr(boxes_b),
           L.ptr(ans_iou), L.stream_ptr())
    return 1


def _nms(boxes, keep, thresh, normal):
    L.require_cuda(boxes)
    if keep.is_cuda or keep.dtype.is_floating_point or keep.element_size() != 8 or not keep.is_contiguous():
        raise L.Det6dError("keep must be a contiguous CPU int64 tensor (iou3d_nms_utils.py:97)")
    if keep.numel() < boxes.shape[0]:
        raise L.Det6dError("keep is shorter than the number of boxes")
    return L.call("det6d_nms_to_host", boxes.shape[0], L.ptr(boxes), float(thresh),
                  ctypes.c_void_p(keep.data_ptr()), normal, L.stream_ptr())


def nms_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(boxes, keep, nms_overlap_thresh, 0)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(boxes, keep, nms_overlap_thresh, 1)


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """iou3d_cpu.cpp:232-252: CPU float32 tensors (N,7), (M,7) -> ans_iou (N,M) filled in place; returns 1"""
    for t in (boxes_a, boxes_b, ans_iou):
        if t.is_cuda or t.dtype != ans_iou.dtype or t.element_size() != 4 or not t.is_contiguous():
            raise L.Det6dError("boxes_iou_bev_cpu needs contiguous CPU float32 tensors (iou3d_cpu.cpp:237-244)")
    if tuple(ans_iou.shape) != (boxes_a.shape[0], boxes_b.shape[0]):
        raise L.Det6dError("ans_iou must be (N, M)")
    L.call("det6d_boxes_iou_bev_cpu", boxes_a.shape[0], ctypes.c_void_p(boxes_a.data_ptr()), boxes_b.shape[0],
           ctypes.c_void_p(boxes_b.data_ptr()), ctypes.c_void_p(ans_iou.data_ptr()))
    return 1
