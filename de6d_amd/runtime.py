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


def mlp_flops_per_scene(model, n_points):
    """ALGORITHMIC flops of every pointwise layer for one scene: 2 * rows * Cin * Cout with the
    true channel counts (no padding), rows = centres x nsample for the grouped MLPs
    (22.583 GFLOP/scene for kitti_models/det6d_car.yaml at 16384 points)."""
    def seq_flops(seq, rows):
        return sum(2.0 * rows * m.in_channels * m.out_channels for m in seq
                   if isinstance(m, (torch.nn.Conv1d, torch.nn.Conv2d)))

    total = 0.0
    for sa in model.backbone_3d.SA_modules:
        m = sum(sa.npoint_list)
        for ns, seq in zip(sa.nsamples, sa.mlps):
            total += seq_flops(seq, m * ns)
        if sa.aggregation_mlp is not None:
            total += seq_flops(sa.aggregation_mlp, m)
        if sa.confidence_mlp is not None:
            total += seq_flops(sa.confidence_mlp, m)
    head = model.point_head
    lo, hi = head.model_cfg.SAMPLE_RANGE
    p = hi - lo
    total += seq_flops(head.vote_layers, p)
    for ns, seq in zip(head.SA_module.nsamples, head.SA_module.mlps):
        total += seq_flops(seq, p * ns)
    total += seq_flops(head.shared_fc_layer, p) + seq_flops(head.cls_layers, p) + seq_flops(head.reg_layers, p)
    return total


class GraphedDet6D(object):
    """One Det6D pass (backbone -> head -> fused post-processing) captured into a hipGraph on its
    own HIP stream: ~130 kernel launches replay with a single host call, so several batches can be
    kept in flight from one Python thread (launch() is asynchronous, finalize() waits).

    The captured graph reads `self.points` (static input, (B*N, 1+3+C)); pass a tensor to launch()
    to have it copied in first, or write into `self.points` yourself."""

    def __init__(self, model, batch_size, n_points, point_width=5, points=None, warmup=2):
        from .ops import fused
        self.model = model
        self.batch_size = batch_size
        self.stream = torch.cuda.Stream()
        self.points = points if points is not None else torch.zeros(
            (batch_size * n_points, point_width), dtype=torch.float32, device='cuda')
        pp = model.model_cfg.POST_PROCESSING
        nms = pp.NMS_CONFIG
        if nms.MULTI_CLASSES_NMS or nms.NMS_TYPE != 'nms_gpu':
            raise NotImplementedError('graph capture needs the fused class-agnostic nms_gpu post-processing')

        def body():
            bd = {'batch_size': batch_size, 'points': self.points}
            for module in model.module_list:
                bd = module(bd)
            out = fused.postprocess(bd['batch_cls_preds'].contiguous(), bd['batch_box_preds'].contiguous(),
                                    batch_size, pp.SCORE_THRESH, nms.NMS_PRE_MAXSIZE, nms.NMS_POST_MAXSIZE,
                                    nms.NMS_THRESH)
            return bd, out

        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(self.stream):
            for _ in range(warmup):
                body()
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=self.stream):
            self.batch_dict, (self.boxes, self.scores, self.labels, self.index, self.count) = body()
        self.count_host = torch.empty(self.count.shape, dtype=self.count.dtype, pin_memory=True)
        self.done = torch.cuda.Event()

    def launch(self, points=None):
        with torch.cuda.stream(self.stream):
            if points is not None and points.data_ptr() != self.points.data_ptr():
                self.points.copy_(points, non_blocking=True)
            self.graph.replay()
            self.count_host.copy_(self.count, non_blocking=True)
            self.done.record()
        return self

    def finalize(self):
        """pred_dicts of the last launch (views into the graph's static outputs)"""
        self.done.synchronize()
        return [{'pred_boxes': self.boxes[i, :k], 'pred_scores': self.scores[i, :k],
                 'pred_labels': self.labels[i, :k].long()} for i, k in enumerate(self.count_host.tolist())]
