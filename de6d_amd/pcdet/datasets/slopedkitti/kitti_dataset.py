"""SlopedKITTI result writer: `KittiDataset.generate_prediction_dicts` plus the `pitch` / `roll` fields and
the two extra label-file columns of core/pcdet/datasets/slopedkitti/kitti_dataset.py:299-379."""
from ..kitti.kitti_dataset import KittiDataset


class SlopedKittiDataset(KittiDataset):
    EXTRA_FIELDS = ('pitch', 'roll')
    EVAL_FUNCTION = 'get_slopedkitti_eval_result'
