"""class_agnostic_nms of core/pcdet/models/model_utils/model_nms_utils.py:6-25."""
import torch

from ...ops.iou3d_nms import iou3d_nms_utils


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    src_box_scores = box_scores
    if score_thresh is not None:
        scores_mask = (box_scores >= score_thresh)
        box_scores = box_scores[scores_mask]
        box_preds = box_preds[scores_mask]
    selected = []
    if box_scores.shape[0] > 0:
        k = min(nms_config.NMS_PRE_MAXSIZE, box_scores.shape[0])
        order = torch.sort(box_scores, dim=0, descending=True, stable=True)[1][:k]  # == topk, ties by index
        keep_idx, _ = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)(
            box_preds[order][:, 0:7], box_scores[order], nms_config.NMS_THRESH, **nms_config)
        selected = order[keep_idx[:nms_config.NMS_POST_MAXSIZE]]
    if score_thresh is not None:
        original_idxs = scores_mask.nonzero().view(-1)
        selected = original_idxs[selected]
    return selected, src_box_scores[selected]
