"""Class-agnostic NMS helper with the signature of
core/pcdet/models/model_utils/model_nms_utils.py:6-25 (generic per-scene route; the fused kernel path
in Detector3DTemplate.post_processing_async does not go through here)."""
import torch

from ...ops.iou3d_nms import iou3d_nms_utils


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """-> (indices into the unfiltered inputs, their scores).  Candidates are the boxes with
    score >= score_thresh, the best NMS_PRE_MAXSIZE of them in stable descending order; survivors
    are cut to NMS_POST_MAXSIZE."""
    all_scores = box_scores
    kept_src = None
    if score_thresh is not None:
        kept_src = torch.nonzero(box_scores >= score_thresh).view(-1)
        box_scores, box_preds = box_scores[kept_src], box_preds[kept_src]
    picked = box_scores.new_zeros((0,), dtype=torch.long)
    n_cand = int(box_scores.shape[0])
    if n_cand:
        top = torch.sort(box_scores, dim=0, descending=True, stable=True)[1][:min(nms_config.NMS_PRE_MAXSIZE, n_cand)]
        nms_fn = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)
        keep, _ = nms_fn(box_preds[top][:, :7], box_scores[top], nms_config.NMS_THRESH, **nms_config)
        picked = top[keep[:nms_config.NMS_POST_MAXSIZE]]
    if kept_src is not None:
        picked = kept_src[picked]
    return picked, all_scores[picked]
