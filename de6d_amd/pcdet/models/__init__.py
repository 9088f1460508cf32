"""build_network / load_data_to_gpu of core/pcdet/models/__init__.py:16-34."""
import numpy as np
import torch

from .detectors import build_detector


def build_network(model_cfg, num_class, dataset):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def load_data_to_gpu(batch_dict):
    for key, val in batch_dict.items():
        if not isinstance(val, np.ndarray) or key in ('frame_id', 'metadata', 'calib'):
            continue
        if key == 'image_shape':
            batch_dict[key] = torch.from_numpy(val).int().cuda()
        else:
            batch_dict[key] = torch.from_numpy(val).float().cuda()
