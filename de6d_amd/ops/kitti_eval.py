"""Device backend of the KITTI / SlopedKITTI evaluator (include/det6d_ops.h, section "KITTI evaluator").

`DeviceEvalBackend` keeps the ragged split (offsets, boxes per metric, scores, alphas) in HBM and exposes
the three steps the evaluator needs: all per-frame overlap blocks of a metric in one launch, pass A
(scores of the true positives, one thread per frame) and pass B (tp / fp / fn / similarity per score
threshold, one thread per frame and threshold, reduced over frames in frame order)."""
import ctypes

import numpy as np
import torch

from .. import _lib as L


def _dev(a, dtype):
    a = np.ascontiguousarray(a, dtype)
    if a.size == 0:                      # keep every pointer non-null (a split may contain no detections at all)
        a = np.zeros((1,) + a.shape[1:], dtype)
    return torch.from_numpy(a).cuda()


class DeviceEvalBackend(object):
    def __init__(self, layout):
        """layout: the `SplitLayout` of pcdet/datasets/kitti/kitti_object_eval_python/eval.py (host NumPy)"""
        self.n_frames = layout.n_frames
        self.total_dt, self.total_gt = int(layout.dt_off[-1]), int(layout.gt_off[-1])
        self.n_pairs = int(layout.pair_off[-1])
        self.dt_f32 = int(layout.dt_f32)
        self.dt_off, self.gt_off = _dev(layout.dt_off, np.int32), _dev(layout.gt_off, np.int32)
        self.pair_off = _dev(layout.pair_off, np.int64)
        self.gt_alpha = _dev(layout.gt_alpha, np.float64)
        self.dt_bbox, self.dt_alpha = _dev(layout.dt_boxes[0], np.float64), _dev(layout.dt_alpha, np.float64)
        self.dt_score = _dev(layout.dt_score, np.float64)
        self._boxes = {m: (_dev(layout.dt_boxes[m], np.float64), _dev(layout.gt_boxes[m], np.float64)) for m in layout.dt_boxes}
        self._overlaps = {}

    def overlaps(self, metric):
        if metric not in self._overlaps:
            out = torch.zeros((max(self.n_pairs, 1),), dtype=torch.float64, device='cuda')
            dt, gt = self._boxes[metric]
            L.call("det6d_eval_overlaps", metric, self.n_frames, L.ptr(self.dt_off), L.ptr(self.gt_off), L.ptr(self.pair_off),
                   ctypes.c_int64(self.n_pairs), L.ptr(dt), L.ptr(gt), self.dt_f32, L.ptr(out), L.stream_ptr())
            self._overlaps[metric] = out
        return self._overlaps[metric]

    def overlaps_host(self, metric):
        return self.overlaps(metric).cpu().numpy()[:self.n_pairs]

    def _args(self, metric, ignored_gt, ignored_dt, dc_off, dc_bbox, min_overlap, n_thresh, compute_aos):
        keep = dict(ign_gt=_dev(ignored_gt, np.int32), ign_dt=_dev(ignored_dt, np.int32), dc_off=_dev(dc_off, np.int32),
                    dc=_dev(dc_bbox.reshape(-1, 4) if len(dc_bbox) else np.zeros((1, 4)), np.float64),
                    ws=torch.empty((max(1, self.total_dt * max(1, n_thresh)),), dtype=torch.uint8, device='cuda'))
        a = L.EvalMatchArgs()
        a.n_frames, a.n_thresh, a.metric, a.compute_aos, a.dt_f32 = self.n_frames, n_thresh, metric, int(compute_aos), self.dt_f32
        a.min_overlap = float(min_overlap)
        for name, t in (("dt_off", self.dt_off), ("gt_off", self.gt_off), ("dc_off", keep['dc_off']), ("pair_off", self.pair_off),
                        ("overlaps", self.overlaps(metric)), ("gt_alpha", self.gt_alpha), ("dt_bbox", self.dt_bbox),
                        ("dt_alpha", self.dt_alpha), ("dt_score", self.dt_score), ("ignored_gt", keep['ign_gt']),
                        ("ignored_dt", keep['ign_dt']), ("dc_bbox", keep['dc']), ("workspace", keep['ws'])):
            setattr(a, name, t.data_ptr())
        return a, keep

    def pass_a(self, metric, ignored_gt, ignored_dt, dc_off, dc_bbox, min_overlap, want_gt_of_tp=False):
        a, keep = self._args(metric, ignored_gt, ignored_dt, dc_off, dc_bbox, min_overlap, 0, False)
        tp_scores = torch.zeros((max(self.total_gt, 1),), dtype=torch.float64, device='cuda')
        tp_count = torch.zeros((max(self.n_frames, 1),), dtype=torch.int32, device='cuda')
        gt_of_tp = torch.full((max(self.total_dt, 1),), -1, dtype=torch.int32, device='cuda') if want_gt_of_tp else None
        a.tp_scores, a.tp_count = tp_scores.data_ptr(), tp_count.data_ptr()
        a.gt_of_tp = gt_of_tp.data_ptr() if want_gt_of_tp else None
        L.call("det6d_eval_match", ctypes.byref(a), L.stream_ptr())
        return (tp_scores.cpu().numpy()[:self.total_gt], tp_count.cpu().numpy()[:self.n_frames],
                gt_of_tp.cpu().numpy()[:self.total_dt] if want_gt_of_tp else None)

    def pass_b(self, metric, ignored_gt, ignored_dt, dc_off, dc_bbox, min_overlap, thresholds, compute_aos):
        n_thresh = len(thresholds)
        if n_thresh == 0:
            return np.zeros((0, 4))
        a, keep = self._args(metric, ignored_gt, ignored_dt, dc_off, dc_bbox, min_overlap, n_thresh, compute_aos)
        thr = _dev(thresholds, np.float64)
        stats = torch.zeros((max(self.n_frames, 1) * n_thresh * 4,), dtype=torch.float64, device='cuda')
        pr = torch.zeros((n_thresh, 4), dtype=torch.float64, device='cuda')
        a.thresholds, a.stats = thr.data_ptr(), stats.data_ptr()
        L.call("det6d_eval_match", ctypes.byref(a), L.stream_ptr())
        L.call("det6d_eval_reduce", self.n_frames, n_thresh, L.ptr(stats), L.ptr(pr), L.stream_ptr())
        return pr.cpu().numpy()
