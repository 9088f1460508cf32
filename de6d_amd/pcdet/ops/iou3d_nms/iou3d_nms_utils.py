"""Python API of core/pcdet/ops/iou3d_nms/iou3d_nms_utils.py:12-116 over libdet6d_hip."""
import torch

from ...ops_backend import fused, iou3d_nms_hip as iou3d_nms_cuda


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """(N,7),(M,7) CPU tensors or numpy arrays -> rotated BEV IoU (N,M) of the same kind (iou3d_nms_utils.py:12-29)"""
    import numpy as np
    if isinstance(boxes_a, np.ndarray):            # common_utils.check_numpy_to_torch (common_utils.py:15-18): .float()
        boxes_a = torch.from_numpy(boxes_a).float()
    is_numpy = isinstance(boxes_b, np.ndarray)     # the reference keeps the flag of boxes_b (iou3d_nms_utils.py:21-22)
    if is_numpy:
        boxes_b = torch.from_numpy(boxes_b).float()
    assert not (boxes_a.is_cuda or boxes_b.is_cuda), 'Only support CPU tensors'
    assert boxes_a.shape[1] == 7 and boxes_b.shape[1] == 7
    ans_iou = boxes_a.new_zeros(torch.Size((boxes_a.shape[0], boxes_b.shape[0])))
    iou3d_nms_cuda.boxes_iou_bev_cpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou.numpy() if is_numpy else ans_iou


def boxes_iou_bev(boxes_a, boxes_b):
    """(N,7),(M,7) device tensors -> rotated BEV IoU (N,M)"""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    ans_iou = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """3-D IoU = BEV overlap x height overlap / union volume (iou3d_nms_utils.py:48-81)"""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_top = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_bot = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_top = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    b_bot = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_overlap_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), overlaps_bev)
    overlaps_h = torch.clamp(torch.min(a_top, b_top) - torch.max(a_bot, b_bot), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def _nms(boxes, scores, thresh, pre_maxsize, normal):
    assert boxes.shape[1] == 7
    # stable: equal scores keep their original order (the reference's torch.sort is unstable)
    order = torch.sort(scores, dim=0, descending=True, stable=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous()
    keep, num = fused.nms_device(boxes, thresh, normal=normal)
    return order[keep[:int(num.item())]].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    return _nms(boxes, scores, thresh, pre_maxsize, False)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    return _nms(boxes, scores, thresh, None, True)
