"""KITTI result writer of the hot path's output side.

Mirror of `KittiDataset.generate_prediction_dicts`
(core/pcdet/datasets/kitti/kitti_dataset.py:277-351): same arguments, same annotation dictionaries,
same label-file lines.  The reference converts every frame on the host (three `.cpu()` pulls per
frame, NumPy box/camera math); here the whole batch is converted by one kernel launch
(det6d_kitti_annos) and pulled with one device-to-host copy.

`evaluation` runs the AP evaluator on the device (kitti_object_eval_python/eval.py).
Loading KITTI from disk and info / database generation are not part of this path.
"""
import numpy as np
import torch

from ....ops import fused

_FIELDS = ('name', 'truncated', 'occluded', 'alpha', 'rotation_y', 'score')


def _image_shape(batch_dict, index):
    shape = batch_dict['image_shape'][index]
    return shape.cpu().numpy() if torch.is_tensor(shape) else np.asarray(shape)


def convert_batch(batch_dict, pred_dicts):
    """one launch + one D2H for the whole batch -> per-frame (annos (K,12), boxes (K,C), scores, labels)"""
    counts = [int(p['pred_scores'].shape[0]) for p in pred_dicts]
    total = sum(counts)
    if total == 0:
        return [None] * len(pred_dicts)
    dev = pred_dicts[0]['pred_boxes'].device
    boxes = torch.cat([p['pred_boxes'] for p in pred_dicts], 0).float().contiguous()
    calib = np.stack([batch_dict['calib'][i].packed(_image_shape(batch_dict, i)) for i in range(len(pred_dicts))])
    scene_of = np.repeat(np.arange(len(pred_dicts), dtype=np.int32), counts)
    annos = fused.kitti_annos(boxes, torch.from_numpy(scene_of).to(dev), torch.from_numpy(calib).to(dev))
    scores = torch.cat([p['pred_scores'] for p in pred_dicts], 0).float()
    labels = torch.cat([p['pred_labels'] for p in pred_dicts], 0)
    packed = torch.cat([annos, boxes, scores[:, None], labels.float()[:, None]], 1).cpu().numpy()
    out, lo, ncol = [], 0, boxes.shape[1]
    for k in counts:
        rows = packed[lo:lo + k]
        out.append(None if k == 0 else (rows[:, :12], rows[:, 12:12 + ncol], rows[:, 12 + ncol],
                                       rows[:, 13 + ncol].astype(np.int64)))
        lo += k
    return out


class KittiDataset(object):
    EXTRA_FIELDS = ()          # SlopedKittiDataset adds ('pitch', 'roll')

    @classmethod
    def empty_prediction(cls, num_samples):
        d = {k: np.zeros(num_samples) for k in _FIELDS + cls.EXTRA_FIELDS}
        d.update(bbox=np.zeros([num_samples, 4]), dimensions=np.zeros([num_samples, 3]),
                 location=np.zeros([num_samples, 3]), boxes_lidar=np.zeros([num_samples, 7]))
        return d

    @classmethod
    def frame_prediction(cls, converted, class_names):
        if converted is None:
            return cls.empty_prediction(0)
        annos, boxes, scores, labels = converted
        d = cls.empty_prediction(len(scores))
        d['name'] = np.array(class_names)[labels - 1]
        d['alpha'] = annos[:, 11]
        d['bbox'] = annos[:, 7:11]
        d['dimensions'] = annos[:, 3:6]     # l, h, w in the camera frame
        d['location'] = annos[:, 0:3]
        d['rotation_y'] = annos[:, 6]
        for j, key in enumerate(cls.EXTRA_FIELDS):
            if boxes.shape[1] >= 9:
                d[key] = boxes[:, 7 + j]
        d['score'] = scores
        d['boxes_lidar'] = boxes
        return d

    @classmethod
    def label_lines(cls, d):
        """'<name> -1 -1 alpha x1 y1 x2 y2 h w l x y z ry [pitch roll] score' with %.4f fields"""
        lines = []
        for i in range(len(d['bbox'])):
            dims, vals = d['dimensions'][i], [d['alpha'][i]]
            vals += list(d['bbox'][i]) + [dims[1], dims[2], dims[0]] + list(d['location'][i]) + [d['rotation_y'][i]]
            vals += [d[key][i] for key in cls.EXTRA_FIELDS]
            vals.append(d['score'][i])
            lines.append('%s -1 -1 %s' % (d['name'][i], ' '.join('%.4f' % v for v in vals)))
        return lines

    @classmethod
    def generate_prediction_dicts(cls, batch_dict, pred_dicts, class_names, output_path=None):
        """
        Args:
            batch_dict: frame_id, calib (list of Calibration), image_shape (B, 2)
            pred_dicts: list of {pred_boxes (N, 7|9), pred_scores (N), pred_labels (N)} device tensors
            class_names, output_path: as in the reference
        Returns: list of annotation dicts (one per frame), label files written if output_path is given
        """
        annos = []
        for index, converted in enumerate(convert_batch(batch_dict, pred_dicts)):
            single = cls.frame_prediction(converted, class_names)
            single['frame_id'] = batch_dict['frame_id'][index]
            annos.append(single)
            if output_path is not None:
                with open(output_path / ('%s.txt' % single['frame_id']), 'w') as f:
                    for line in cls.label_lines(single):
                        print(line, file=f)
        return annos

    EVAL_FUNCTION = 'get_official_eval_result'

    def evaluation(self, det_annos, class_names, gt_annos=None, **kwargs):
        """AP report + dictionary (kitti_dataset.py:353-363 / slopedkitti :385-394) through the device evaluator.
        Ground truth comes from `self.kitti_infos[i]['annos']` like in the reference, or from `gt_annos`."""
        import copy
        from .kitti_object_eval_python import eval as kitti_eval
        if gt_annos is None:
            infos = getattr(self, 'kitti_infos', None)
            if not infos or 'annos' not in infos[0].keys():
                return None, {}
            gt_annos = [info['annos'] for info in infos]
        return getattr(kitti_eval, self.EVAL_FUNCTION)(copy.deepcopy(gt_annos), copy.deepcopy(det_annos), class_names)
