"""Small host-side helpers shared by bench.py, __graft_entry__.py and the tests: build a Det6D
model from a YAML the way tools/test.py does (core/tools/test.py:21-65), seeded weights, and a
dataset stand-in exposing the attributes Detector3DTemplate.build_networks reads
(core/pcdet/models/detectors/detector3d_template.py:36-44)."""
import os
import time

import numpy as np
import torch

from . import _lib as _L
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


#: Farthest point sampling is ONE 1024-thread workgroup per scene running a 4095-round latency chain.  Inside a
#: stream of GEMM workgroups at equal queue priority such a workgroup waits 10-20 ms for a CU with enough free
#: registers / wave slots (scripts/experiments/gpu_cumask.py: 3.6 ms alone, 19 ms beside four GEMM streams, 5.5 ms from a
#: high-priority queue), which kept 22 passes in flight, most of them waiting for a sampler.  HIP gives
#: high-priority streams only FOUR hardware queues (scripts/experiments/gpu_prio_queues.py), so instead the samplers
#: that depend on nothing but the input cloud (hoist_plan) are cut out of the captured passes and launched ONCE for a
#: group of passes on a sampler stream, AHEAD of the passes' GEMM stage (Det6DGroup); everything else replays as graph
#: segments on the main streams.
SAMPLER_GROUP = 4   # passes per group in bench.py (--group)


def hoist_plan(sa_modules, n_points):
    """Which samplers of the backbone depend on NOTHING but the input cloud?  The first layer's d-fps does, and so does
    every later d-fps whose range lies inside the picks of such a sampler of the layer before (Det6D: 4096 of the input
    -> d-fps 512 of those 4096 -> d-fps 256 of those 512; the s-fps halves need the confidence scores of the layer
    before).  Returns the launch list of a group's stage 1, in order:
      dict(layer, j, src=(layer-1, j') | None, lo, hi (relative to src), m, offset (column in the layer's index buffer),
           bias (added to every pick on top of lo: start of src inside the previous layer's output), feeds=bool)"""
    plan, outs = [], {}
    n_in = n_points
    for layer, sa in enumerate(sa_modules):
        jobs = list(zip(sa.sample_range_list, sa.sample_method_list, sa.npoint_list))
        offsets = [sum(sa.npoint_list[:i]) for i in range(len(jobs))]
        for j, ((lo, hi), method, npoint) in enumerate(jobs):
            hi = n_in if hi == -1 else hi
            if method != 'd-fps':
                continue
            if layer == 0:
                if not (len(jobs) == 1 and lo == 0 and hi == n_in):
                    raise NotImplementedError("grouped first sampler: expected one d-fps over the whole input cloud")
                plan.append(dict(layer=0, j=0, src=None, lo=0, hi=n_in, m=npoint, offset=0, bias=0, feeds=False))
                outs[(0, 0)] = (0, npoint)
                continue
            for (pl, pj), (poff, pn) in list(outs.items()):
                if pl == layer - 1 and poff <= lo and hi <= poff + pn:
                    plan.append(dict(layer=layer, j=j, src=(pl, pj), lo=lo - poff, hi=hi - poff, m=npoint, offset=offsets[j],
                                     bias=poff, feeds=False))
                    outs[(layer, j)] = (offsets[j], npoint)
                    break
        n_in = sum(sa.npoint_list)
    for step in plan:
        if step['src'] is not None:
            for other in plan:
                if (other['layer'], other['j']) == step['src']:
                    other['feeds'] = True
    return plan


class _SegmentCapture(object):
    """capture controller of a pass that belongs to a Det6DGroup: graph segments on `main`, cut where the first SA layer
    samples (the group launches its hoisted samplers for all its passes between segment 0 = pack and the rest); the SA
    layers take their index buffers from the group and skip the samplers the group launches (`front`)."""

    def __init__(self, main, pool, front):
        self.main, self.pool = main, pool
        self.front = front               # dict(rows, xyz, idx={layer: (B, M) slice}, hoisted={(layer, j)})
        self.segments = []               # [graph, ...]
        self.pack_out = (front['rows'], front['xyz'])
        self._graph = self._ctx = None
        self._layer = -1

    def begin(self):
        self._graph = torch.cuda.CUDAGraph()
        self._ctx = torch.cuda.stream(self.main)
        self._ctx.__enter__()
        self._graph.capture_begin(pool=self.pool)

    def end(self):
        self._graph.capture_end()
        self._ctx.__exit__(None, None, None)
        self.segments.append(self._graph)
        self._graph = self._ctx = None

    def next_layer(self):
        self._layer += 1
        if self._layer == 0:             # segment 0 ends here; the group's stage 1 runs between the segments
            self.end()
            self.begin()
        return self._layer

    def index_buffer(self, layer, b, m):
        buf = self.front['idx'].get(layer)
        if buf is not None:
            assert tuple(buf.shape) == (b, m)
        return buf

    def hoisted(self, layer, j):
        return (layer, j) in self.front['hoisted']

    def after_samplers(self, layer, xyz, idx):
        pass                             # the group has launched them ahead of the segments


class _InlineHoist(object):
    """controller of a SINGLE-graph pass (GraphedDet6D without a group): the first layer's sampler runs in place, then the
    rest of the input-only chain (hoist_plan: the d-fps halves of the later layers) is forked onto a side stream — a
    parallel branch of the captured graph — and joined where a later layer needs its picks, so those samplers overlap
    the first layer's MLPs and the score-weighted samplers instead of queueing behind them (latency only)."""

    def __init__(self, model, batch_size, n_points):
        from .ops import fused
        sa = list(model.backbone_3d.SA_modules)
        try:
            self.plan = hoist_plan(sa, n_points)
        except NotImplementedError:
            self.plan = []
        self.pack_out = None
        self._fused = fused
        self.side = torch.cuda.Stream()
        self.idx, self.ctr, self.ws, self.events = {}, {}, {}, {}
        dev = 'cuda'
        for step in self.plan[1:]:
            layer = step['layer']
            if layer not in self.idx:
                self.idx[layer] = torch.empty((batch_size, sum(sa[layer].npoint_list)), dtype=torch.int32, device=dev)
                self.events[layer] = torch.cuda.Event()
            self.ws[(layer, step['j'])] = fused.fps_workspace(batch_size, step['hi'] - step['lo'], dev)
        for step in self.plan:
            if step['feeds']:
                self.ctr[(step['layer'], step['j'])] = torch.empty((batch_size, step['m'], 3), dtype=torch.float32, device=dev)
        self._later = {(s['layer'], s['j']) for s in self.plan[1:]}
        self._layer = -1
        self.status_words = []       # error words of samplers that can fail after their launch (fused.fps_fused appends)

    def reset(self):
        self._layer = -1
        del self.status_words[:]

    def next_layer(self):
        self._layer += 1
        return self._layer

    def index_buffer(self, layer, b, m):
        return self.idx.get(layer)

    def hoisted(self, layer, j):
        return (layer, j) in self._later

    def after_samplers(self, layer, xyz, idx):
        F = self._fused
        main = torch.cuda.current_stream()
        if layer == 0 and self._later:
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                first = self.plan[0]
                prev_xyz, prev_idx = xyz, idx
                for step in self.plan:
                    key = (step['layer'], step['j'])
                    if step is not first:
                        src = self.ctr[step['src']]
                        F.fps_fused(src, step['lo'], step['hi'], step['m'], None, 1.0, self.idx[step['layer']], step['offset'],
                                    temp=self.ws[key], idx_bias=step['bias'])
                        self.events[step['layer']].record(self.side)
                        prev_xyz, prev_idx = src, self.idx[step['layer']]
                    if step['feeds']:
                        F.gather_centres(prev_xyz, prev_idx[:, step['offset']:step['offset'] + step['m']], out=self.ctr[key],
                                         idx_bias=-step['bias'])
        elif layer in self.events:
            main.wait_event(self.events[layer])
            if layer == max(self.events):
                main.wait_stream(self.side)          # every forked branch joins the capturing stream


class GraphedDet6D(object):
    """One Det6D pass (backbone -> head -> fused post-processing) captured into a hipGraph on its
    own HIP stream: ~130 kernel launches replay with a single host call, so several batches can be
    kept in flight from one Python thread (launch() is asynchronous, finalize() waits).

    The captured graph reads `self.points` (static input, (B*N, 1+3+C)); pass a tensor to launch()
    to have it copied in first, or write into `self.points` yourself."""

    def __init__(self, model, batch_size, n_points, point_width=5, points=None, warmup=2, front=None, stream=None):
        """front = dict(rows, xyz, idx, hoisted) of a Det6DGroup: the pass packs its points into the group's slices and
        takes the picks of the hoisted samplers from the group's index buffers (the group launches them for all its passes)"""
        from .ops import fused
        self.model = model
        self.batch_size = batch_size
        self.stream = stream if stream is not None else torch.cuda.Stream()
        self.points = points if points is not None else torch.zeros(
            (batch_size * n_points, point_width), dtype=torch.float32, device='cuda')
        pp = model.model_cfg.POST_PROCESSING
        nms = pp.NMS_CONFIG
        if nms.MULTI_CLASSES_NMS or nms.NMS_TYPE != 'nms_gpu':
            raise NotImplementedError('graph capture needs the fused class-agnostic nms_gpu post-processing')

        inline = None
        forked = _L.experiment_switch('DET6D_FORKED_SAMPLERS') is not None      # the samplers of a layer on forked streams
        if forked and front is not None:
            raise RuntimeError("DET6D_FORKED_SAMPLERS=1 cannot be combined with a Det6DGroup (its passes take the hoisted "
                               "samplers' picks from the group's buffers)")
        if front is None and _L.experiment_switch('DET6D_NO_HOIST') is None and not forked:
            inline = _InlineHoist(model, batch_size, n_points)

        def body():
            if inline is not None:
                inline.reset()
            bd = {'batch_size': batch_size, 'points': self.points}
            for module in model.module_list:
                bd = module(bd)
            boxes, scores, labels, index, count = fused.postprocess(
                bd['batch_cls_preds'].contiguous(), bd['batch_box_preds'].contiguous(), batch_size, pp.SCORE_THRESH,
                nms.NMS_PRE_MAXSIZE, nms.NMS_POST_MAXSIZE, nms.NMS_THRESH)
            # pred_labels are int64 in the reference (detector3d_template.py:239): converted once, inside the graph
            return bd, (boxes, scores, labels.long(), index, count)

        self.stream.wait_stream(torch.cuda.current_stream())
        fused.SAMPLER_SEGMENTS = inline
        try:
            with torch.no_grad(), torch.cuda.stream(self.stream):
                for _ in range(warmup):
                    body()
        finally:
            fused.SAMPLER_SEGMENTS = None
        torch.cuda.synchronize()
        self.segments = None
        if front is not None:
            torch.cuda.synchronize()
            ctl = _SegmentCapture(self.stream, torch.cuda.graph_pool_handle(), front=front)
            fused.SAMPLER_SEGMENTS = ctl
            try:
                with torch.no_grad():
                    ctl.begin()
                    self.batch_dict, (self.boxes, self.scores, self.labels, self.index, self.count) = body()
                    ctl.end()
            finally:
                fused.SAMPLER_SEGMENTS = None
            self.segments = ctl.segments
        else:
            self.graph = torch.cuda.CUDAGraph()
            fused.SAMPLER_SEGMENTS = inline
            try:
                with torch.no_grad(), torch.cuda.graph(self.graph, stream=self.stream):
                    self.batch_dict, (self.boxes, self.scores, self.labels, self.index, self.count) = body()
            finally:
                fused.SAMPLER_SEGMENTS = None
        # The result block (B, P, .) is static across replays and its slots past a scene's count are never written: clear it
        # once, so that a consumer that converts whole blocks (datasets/kitti: convert_batch) reads zeros or earlier finite
        # detections there, never uninitialised memory
        with torch.cuda.stream(self.stream):
            for t in (self.boxes, self.scores, self.labels):
                t.zero_()
        self.stream.synchronize()
        self.count_host = torch.empty(self.count.shape, dtype=self.count.dtype, pin_memory=True)
        self.done = torch.cuda.Event()
        self._weights_version = getattr(model, 'weights_version', 0)
        # samplers of this pass that can fail after their launch (the cooperative 32768 / 65536-point sampler): their
        # sticky error words travel to pinned memory next to the counts and are looked at in finalize()
        self._status_words = list(inline.status_words) if inline is not None else []
        self._status_host = torch.zeros((max(1, len(self._status_words)),), dtype=torch.int32, pin_memory=True)
        self._front_group = None      # the Det6DGroup this pass belongs to (its stage-1 samplers' status: see Det6DGroup)
        self._front_gen = -1          # generation of the group launch this pass's last launch_rest() belonged to

    def _relaunched(self):
        lazy = self.batch_dict.get('point_coords_list', None)
        if hasattr(lazy, 'reset'):
            lazy.reset()           # lists built on first access from the centres: stale after a replay

    def _check_weights(self):
        if getattr(self.model, 'weights_version', 0) != self._weights_version:
            raise RuntimeError("the model's weights changed after this pass was captured (load_state_dict / train()): "
                               "its graph still points at the old folded matrices; build a new GraphedDet6D / Det6DGroup")

    def launch_front(self, points=None):
        """segment 0 (everything before the first SA layer's samplers) on the CURRENT stream (the group's sampler stream), once
        the pass's previous launch has finished with the buffers"""
        self._check_weights()
        self._relaunched()
        if GraphedDet6D.stamp_launches:
            self.issued_at = time.perf_counter()     # host time at which this launch of the pass was issued (bench.py)
        torch.cuda.current_stream().wait_event(self.done)
        if callable(points):          # an input producer filling self.points on the current stream (bench.py pipeline leg)
            points(self)
        elif isinstance(points, (list, tuple)):     # the batches a coalesced pass is made of, in order
            at = 0
            for p in points:
                self.points[at:at + p.shape[0]].copy_(p, non_blocking=True)
                at += p.shape[0]
        elif points is not None and points.data_ptr() != self.points.data_ptr():
            self.points.copy_(points, non_blocking=True)
        self.segments[0].replay()

    def launch_rest(self, sampled):
        """the remaining segments once the group's sampler (event `sampled`) has run"""
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(sampled)
            for graph in self.segments[1:]:
                graph.replay()
            self.count_host.copy_(self.count, non_blocking=True)
            self._copy_status()
            self.done.record()
            self._stamp()
        return self

    def _copy_status(self):
        for i, w in enumerate(self._status_words):
            self._status_host[i:i + 1].copy_(w, non_blocking=True)

    def _check_status(self):
        """after `done`: did a sampler of this pass (or of its group's stage 1) give up?  (fps_coop.hip's time-out)"""
        from .ops import fused
        bad = []
        if self._status_words and bool(self._status_host[:len(self._status_words)].any()):
            bad += [w for w, f in zip(self._status_words, self._status_host.tolist()) if f]
        if bad:
            with torch.cuda.stream(self.stream):     # ordered against the next replay of this pass
                for w in bad:
                    w.zero_()
            self._status_host.zero_()
        front_bad = self._front_group is not None and self._front_group.launch_failed(self._front_gen)
        if bad or front_bad:
            raise fused.FpsTimeout("a cooperative farthest-point sampler of this pass gave up waiting for its partner "
                                   "workgroups (include/det6d_ops.h: det6d_fps_fused_status): the pass's detections are invalid")

    #: bench.py sets this to have every launch leave a timing event (`self.stamp`: device-side completion time of the pass)
    stamp_launches = False

    def _stamp(self):
        if GraphedDet6D.stamp_launches:
            self.stamp = torch.cuda.Event(enable_timing=True)
            self.stamp.record()

    def launch(self, points=None):
        self._check_weights()
        self._relaunched()
        if GraphedDet6D.stamp_launches:
            self.issued_at = time.perf_counter()
        with torch.cuda.stream(self.stream):
            if points is not None and points.data_ptr() != self.points.data_ptr():
                self.points.copy_(points, non_blocking=True)
            if self.segments is not None:
                raise RuntimeError("this pass belongs to a Det6DGroup: launch the group")
            self.graph.replay()
            self.count_host.copy_(self.count, non_blocking=True)
            self._copy_status()
            self.done.record()
            self._stamp()
        return self

    #: seconds the host spent blocked in finalize() (all passes): host-bound pipelines show ~0 here
    host_wait_s = 0.0

    def finalize(self):
        """pred_dicts of the last launch (views into the graph's static outputs)"""
        if not self.done.query():
            t0 = time.perf_counter()
            self.done.synchronize()
            GraphedDet6D.host_wait_s += time.perf_counter() - t0
        self._check_status()
        return [{'pred_boxes': self.boxes[i, :k], 'pred_scores': self.scores[i, :k],
                 'pred_labels': self.labels[i, :k]} for i, k in enumerate(self.count_host.tolist())]


class Det6DGroup(object):
    """K captured passes whose INPUT-ONLY samplers (hoist_plan: the first layer's D-FPS and the d-fps chain below it, which
    depend on nothing but the input cloud) run as one launch each over all K x B scenes on a stream of its own.  Two stages:
      launch_front(): pack + the hoisted samplers of the K passes on `sampler_stream` (waits until the passes' previous
                      launch is done with the buffers);
      launch_rest():  the remaining graph segments of every pass on its main stream, after the samplers.
    Issue launch_front() of later groups BEFORE launch_rest() of earlier ones and the samplers (one 1024-thread
    workgroup per scene, a 3.5 ms latency chain that waits 10-20 ms for a free CU beside GEMM traffic) run ahead of
    the GEMM stage instead of blocking its streams.  launch() = both stages back to back."""

    def __init__(self, model, batch_size, n_points, k, sampler_stream, point_width=5, points=None, main_streams=None):
        from .ops import fused
        from .pcdet.ops.pointnet2.pointnet2_batch.pointnet2_modules import rows_ld
        sa_modules = list(model.backbone_3d.SA_modules)
        ld = rows_ld(point_width - 4)
        dev = 'cuda'
        nb = k * batch_size
        self.k, self.batch_size, self.n_points = k, batch_size, n_points
        self.hi = sampler_stream
        self.plan = hoist_plan(sa_modules, n_points)
        if _L.experiment_switch('DET6D_NO_HOIST'):          # only the first layer's sampler ahead of the passes (round-1 behaviour)
            self.plan = self.plan[:1]
            self.plan[0]['feeds'] = False
        self.rows_all = torch.empty((nb, n_points, ld), dtype=torch.float32, device=dev)
        self.xyz_all = torch.empty((nb, n_points, 3), dtype=torch.float32, device=dev)
        self.idx_all, self.ctr_all, self.ws = {}, {}, {}
        for step in self.plan:
            layer = step['layer']
            if layer not in self.idx_all:
                self.idx_all[layer] = torch.empty((nb, sum(sa_modules[layer].npoint_list)), dtype=torch.int32, device=dev)
            if step['feeds']:
                self.ctr_all[(layer, step['j'])] = torch.empty((nb, step['m'], 3), dtype=torch.float32, device=dev)
            self.ws[(layer, step['j'])] = fused.fps_workspace(nb, step['hi'] - step['lo'], dev)
        hoisted = {(s['layer'], s['j']) for s in self.plan}
        self._status_words = [w for w in (fused.fps_status_word(nb, s['hi'] - s['lo'], self.ws[(s['layer'], s['j'])])
                                          for s in self.plan) if w is not None]
        # One row of pinned status words per launch in flight (ring): every pass of a launch looks at ITS launch's row, so
        # the first finalize() that sees a failure does not hide it from the other k-1 passes of that launch.
        self._status_ring = 16
        self._status_host = torch.zeros((self._status_ring, max(1, len(self._status_words))), dtype=torch.int32, pin_memory=True)
        self._gen = 0                 # launches so far
        self._cleared_gen = -1        # last failed launch whose device words have been cleared
        self.runners = []
        for j in range(k):
            sl = slice(j * batch_size, (j + 1) * batch_size)
            own = points[j % len(points)] if isinstance(points, (list, tuple)) else points   # a static input per pass
            front = dict(rows=self.rows_all[sl], xyz=self.xyz_all[sl], idx={l: t[sl] for l, t in self.idx_all.items()},
                         hoisted=hoisted)
            self.runners.append(GraphedDet6D(model, batch_size, n_points, point_width, points=own, front=front,
                                             stream=None if main_streams is None else main_streams[j % len(main_streams)]))
        if self._status_words:
            for r in self.runners:
                r._front_group = self
        self._fused = fused
        self._sampled = torch.cuda.Event()
        self._count = k

    def launch_front(self, points=None, count=None):
        self._count = self.k if count is None else count
        nb = self._count * self.batch_size
        F = self._fused
        with torch.cuda.stream(self.hi):
            for r in self.runners[:self._count]:
                r.launch_front(points)
            for step in self.plan:
                key = (step['layer'], step['j'])
                src = self.xyz_all[:nb] if step['src'] is None else self.ctr_all[step['src']][:nb]
                idx = self.idx_all[step['layer']]
                F.fps_fused(src, step['lo'], step['hi'], step['m'], None, 1.0, idx[:nb], step['offset'], temp=self.ws[key],
                            idx_bias=step['bias'])
                if step['feeds']:     # xyz of these picks: the cloud the next layer's hoisted sampler works on
                    F.gather_centres(src, idx[:nb, step['offset']:step['offset'] + step['m']], out=self.ctr_all[key][:nb],
                                     idx_bias=-step['bias'])
            self._gen += 1
            row = self._status_host[self._gen % self._status_ring]
            for i, w in enumerate(self._status_words):     # sticky error words -> this launch's pinned row, looked at in finalize()
                row[i:i + 1].copy_(w, non_blocking=True)
            self._sampled.record(self.hi)
        return self

    def launch_failed(self, gen):
        """did a hoisted sampler give up in (or before, and unreported until) launch `gen`?  Every pass of that launch gets
        the same answer.  The device words are sticky: they are cleared here, once per failed launch, on the sampler stream
        (ordered against the next stage-1 launch), so launches issued before the clear report the failure too."""
        if gen < 0:
            return False
        if gen <= self._gen - self._status_ring:
            # the pinned row of that launch has been reused by a later one: whether its samplers gave up is no longer known
            raise RuntimeError("Det6DGroup: launch %d was finalized %d launches late (status ring of %d): finalize every pass of "
                               "a group launch before the group has been launched %d more times"
                               % (gen, self._gen - gen, self._status_ring, self._status_ring))
        if not bool(self._status_host[gen % self._status_ring].any()):
            return False
        if gen > self._cleared_gen:
            self._cleared_gen = gen
            with torch.cuda.stream(self.hi):
                for w in self._status_words:
                    w.zero_()
        return True

    def launch_rest(self):
        active = self.runners[:self._count]
        for r in active:
            r._front_gen = self._gen
            r.launch_rest(self._sampled)
        return active

    def launch(self, points=None, count=None):
        return self.launch_front(points, count).launch_rest()


#: Workgroups of the cooperative samplers that may be in flight together, as a fraction of the device's CUs (one 1024-thread
#: workgroup holds a CU).  1.0 is exact, not optimistic: with at most #CUs such workgroups in flight, a launch that is partly
#: resident can hold at most (#CUs - the other launches' sizes) CUs, so every launch can always become fully resident once the
#: kernels that do not wait for anybody have drained; beyond #CUs, partly resident launches can starve each other.
COOP_CU_FRACTION = 1.0


def hip_initialised():
    """has this process initialised the HIP runtime through torch?  (GPU_MAX_HW_QUEUES is read at that moment)"""
    return bool(torch.cuda.is_initialized())


def require_hw_queues(need, allow_aliasing=False):
    """ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable ONCE, when the runtime
    initialises: with fewer queues than streams the passes in flight alias onto the queues and serialise (2 051 instead of
    9 680 scenes/s, bench.py:56-61) — silently.  `import de6d_amd` exports 24 when nothing has touched the GPU yet
    (de6d_amd/__init__.py); a process that initialised HIP first with fewer queues than `need` gets a RuntimeError here instead
    of a 4x slower pipeline, unless it asks for the aliased form (`allow_aliasing=True`: results are identical, only slower)."""
    from . import HW_QUEUES_AT_IMPORT
    have = HW_QUEUES_AT_IMPORT if HW_QUEUES_AT_IMPORT is not None else int(os.environ.get('GPU_MAX_HW_QUEUES', '4'))
    if have < need and not allow_aliasing:
        raise RuntimeError("the pipeline uses %d HIP streams but this process has GPU_MAX_HW_QUEUES=%d hardware queues (the variable "
                           "is read when the HIP runtime initialises): export GPU_MAX_HW_QUEUES=24 — or import de6d_amd — before "
                           "the process touches the GPU, or pass allow_aliasing=True to run on aliased, serialising queues"
                           % (need, have))
    return have


class ScenePipeline(object):
    """The throughput runner: batches of scenes stream through `n_main / group + prefetch` Det6DGroups.
    Stage 1 of a group (pack + the input-only samplers of its `group` passes) is issued `prefetch` groups ahead on one of the
    sampler streams, stage 2 (the captured rest of every pass) on `n_main` main streams; a pass is finalised (its
    detections sliced per scene, the only host sync) when the slot it occupies is needed again.

    merge: consecutive batches coalesced into ONE pass (a pass then holds merge x batch_size scenes).  Every kernel of the
           path works scene by scene (BN in eval mode, per-scene samplers / ball queries / NMS), so a scene's result does not
           depend on what shares its pass — bench.py's self-check and tests/test_timed_path_gpu.py compare coalesced passes
           with one-batch passes bit for bit — but 4x larger launches fill the 256 CUs better (+10 % scenes/s at batch 8).
           A STEP stays one batch of batch_size scenes: run(steps) and on_done count batches.

    points: None — every pass owns a static input buffer (`passes[i].points`, merge x batch_size scenes) the caller fills
            or copies into; a tensor — every pass reads it; a list of tensors — pass i reads points[i % len(points)]
            (each one pass long: see coalesce())."""

    @staticmethod
    def coalesce(batches, merge):
        """resident per-batch inputs -> per-pass inputs: pass i = batches i*merge .. i*merge+merge-1 (cyclic) back to back"""
        if merge <= 1:
            return list(batches)
        n_pass = max(1, (len(batches) + merge - 1) // merge)
        return [torch.cat([batches[(i * merge + j) % len(batches)] for j in range(merge)], 0) for i in range(n_pass)]

    def __init__(self, model, batch_size, n_points, n_main=16, group=4, prefetch=4, sampler_streams=6, points=None,
                 point_width=5, main_streams=None, samplers=None, merge=1, allow_aliasing=False):
        self.merge = max(1, int(merge))
        self.step_scenes = batch_size
        batch_size = batch_size * self.merge
        self.k = max(1, min(group, n_main))
        self.prefetch = prefetch
        from .ops import fused
        n_samplers = len(samplers) if samplers else sampler_streams
        if fused.fps_is_cooperative(n_points):
            # The cooperative sampler of 32768 / 65536-point scenes needs ALL parts of a scene resident at once (they
            # poll each other).  Launches on different streams may be dispatched interleaved, so the launches in flight
            # together must fit the chip with one 1024-thread workgroup per CU (COOP_CU_FRACTION): stage 1 of the
            # groups shares as few sampler streams as that allows (launches on one stream never overlap; inside one
            # launch the parts of a scene are consecutive in dispatch order).
            per_launch = min(group, n_main) * batch_size * (n_points // 16384)
            cus = int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count * COOP_CU_FRACTION)
            n_samplers = min(n_samplers, max(1, cus // max(1, per_launch)))
        # the streams this pipeline will actually use (sampler streams counted AFTER the cooperative trim)
        self.hw_queues = require_hw_queues((len(main_streams) if main_streams else n_main) + n_samplers, allow_aliasing)
        self.main_streams = list(main_streams) if main_streams else [torch.cuda.Stream() for _ in range(n_main)]
        self.sampler_streams = (list(samplers) if samplers else [torch.cuda.Stream() for _ in range(n_samplers)])[:n_samplers]
        n_main = len(self.main_streams)
        self.n_groups = max(1, n_main // self.k) + prefetch
        self.groups = []
        for g in range(self.n_groups):
            own = points
            if isinstance(points, (list, tuple)):
                own = [points[(g * self.k + j) % len(points)] for j in range(self.k)]
            self.groups.append(Det6DGroup(model, batch_size, n_points, self.k, self.sampler_streams[g % len(self.sampler_streams)],
                                          point_width=point_width, points=own,
                                          main_streams=[self.main_streams[(g * self.k + j) % n_main] for j in range(self.k)]))
        self.passes = [r for grp in self.groups for r in grp.runners]
        self.prime()

    def prime(self):
        """every group once through both stages (first launch of each captured graph, code objects, workspaces), so that
        no later run pays for it whatever its length"""
        for grp in self.groups:
            grp.launch()
        for r in self.passes:
            r.finalize()
        torch.cuda.synchronize()

    def run(self, steps, feed=None, on_done=None, headway=0.0):
        """`steps` batches through the pipeline (ceil(steps / merge) passes; the last pass of a stream whose length is not a
        multiple of `merge` runs full and reports its leading batches only); `feed` (host tensor one pass long, list of
        the batches of a pass, or callable(pass)) supplies the input of a pass whose static buffer is not resident already;
        on_done(step, pass, pred_dicts of that batch) is called in step order as passes are finalised.  Returns the number
        of steps finalised (== steps).  `headway` [s] > 0 keeps consecutive GEMM-stage launches at least that far apart
        (headway control: passes that retire together are not re-issued together, so the passes in flight sit at different
        depths of the network instead of marching through it in lock-step); no effect on results."""
        k, n_groups, prefetch, merge, sb = self.k, self.n_groups, self.prefetch, self.merge, self.step_scenes
        t_next = time.perf_counter()
        counts, left = [], (steps + merge - 1) // merge
        while left > 0:
            counts.append(min(k, left))
            left -= counts[-1]
        done, inflight = 0, []

        def finish(active):
            nonlocal done
            for r in active:
                preds = r.finalize()
                for j in range(merge):
                    if done < steps:
                        if on_done is not None:
                            on_done(done, r, preds[j * sb:(j + 1) * sb])
                        done += 1

        for g in range(min(prefetch, len(counts))):
            self.groups[g % n_groups].launch_front(feed, counts[g])
        for g in range(len(counts)):
            if len(inflight) >= n_groups - prefetch:
                finish(inflight.pop(0))
            if g + prefetch < len(counts):
                self.groups[(g + prefetch) % n_groups].launch_front(feed, counts[g + prefetch])
            if headway > 0.0:
                while time.perf_counter() < t_next:
                    time.sleep(5e-5)
                t_next = time.perf_counter() + headway
            inflight.append(self.groups[g % n_groups].launch_rest())
        for active in inflight:
            finish(active)
        return done
