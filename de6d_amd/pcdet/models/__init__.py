"""`build_network` / `load_data_to_gpu`, the two entry points tools/test.py and the ROS node use
(core/pcdet/models/__init__.py:16-34)."""
import numpy as np
import torch

from .detectors import build_detector

_HOST_ONLY_KEYS = frozenset(('frame_id', 'metadata', 'calib'))


def build_network(model_cfg, num_class, dataset):
    """YAML MODEL section -> detector (registry lookup by MODEL.NAME)"""
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def load_data_to_gpu(batch_dict):
    """numpy entries of a collated batch -> device tensors, in place: float32 for everything except
    `image_shape` (int32); bookkeeping keys stay on the host"""
    for key in list(batch_dict.keys()):
        val = batch_dict[key]
        if key in _HOST_ONLY_KEYS or not isinstance(val, np.ndarray):
            continue
        t = torch.from_numpy(val)
        batch_dict[key] = (t.int() if key == 'image_shape' else t.float()).cuda()
