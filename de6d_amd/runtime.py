"""Small host-side helpers shared by bench.py, __graft_entry__.py and the tests: build a Det6D
model from a YAML the way tools/test.py does (core/tools/test.py:21-65), seeded weights, and a
dataset stand-in exposing the attributes Detector3DTemplate.build_networks reads
(core/pcdet/models/detectors/detector3d_template.py:36-44)."""
import os

import numpy as np
import torch

from .pcdet.config import EasyDict, cfg_from_yaml_file
from .pcdet.models import build_network

CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cfgs')


class DatasetStub(object):
    """what build_network needs from a DatasetTemplate (dataset.py:25-36)"""

    def __init__(self, class_names, num_point_features=4, point_cloud_range=(0, -40, -3, 70.4, 40, 1)):
        self.class_names = list(class_names)
        self.point_feature_encoder = EasyDict(num_point_features=num_point_features)
        self.grid_size = None
        self.voxel_size = None
        self.point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        self.depth_downsample_factor = None


def load_config(name_or_path):
    path = name_or_path if os.path.isfile(name_or_path) else os.path.join(CFG_DIR, name_or_path)
    return cfg_from_yaml_file(path, EasyDict())


def randomize_bn_stats(model, seed=4321):
    """non-trivial running statistics so that BN folding is exercised"""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
            m.bias.data.copy_(0.1 * torch.randn(m.bias.shape, generator=g))


def build_model(cfg, seed=1234, device=None, state_dict=None):
    """random-init (seeded) Det6D in eval mode; `state_dict` overrides the weights if given"""
    torch.manual_seed(seed)
    ds = DatasetStub(cfg.CLASS_NAMES, point_cloud_range=cfg.DATA_CONFIG.POINT_CLOUD_RANGE)
    model = build_network(model_cfg=cfg.MODEL, num_class=len(cfg.CLASS_NAMES), dataset=ds)
    with torch.no_grad():
        randomize_bn_stats(model)
    if state_dict is not None:
        model._load_state_dict(state_dict, strict=True)
    model.eval()
    if device is not None:
        model.to(device)
    return model
