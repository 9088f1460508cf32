"""Small host-side helpers shared by bench.py, __graft_entry__.py and the tests: build a Det6D
model from a YAML the way tools/test.py does (core/tools/test.py:21-65), seeded weights, and a
dataset stand-in exposing the attributes Detector3DTemplate.build_networks reads
(core/pcdet/models/detectors/detector3d_template.py:36-44)."""
import os
import time

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


#: Farthest point sampling is ONE 1024-thread workgroup per scene running a 4095-round latency chain.  Inside a
#: stream of GEMM workgroups at equal queue priority such a workgroup waits 10-20 ms for a CU with enough free
#: registers / wave slots (scripts/gpu_cumask.py: 3.6 ms alone, 19 ms beside four GEMM streams, 5.5 ms from a
#: high-priority queue), which kept 22 passes in flight, most of them waiting for a sampler.  HIP gives
#: high-priority streams only FOUR hardware queues (scripts/gpu_prio_queues.py), so instead the first sampler — the
#: one that depends on nothing but the input cloud — is cut out of the captured passes and launched ONCE for a group
#: of passes on a sampler stream, AHEAD of the passes' GEMM stage (Det6DGroup); everything else replays as graph
#: segments on the main streams.
SAMPLER_GROUP = 4   # passes per group in bench.py (--group)


class _SegmentCapture(object):
    """capture controller: graph segments on `main`; the first sampler is left to the group (whose `front` buffers the
    pass packs its input into and reads the sampled indices from), later samplers stay inside the graph"""

    def __init__(self, main, pool, front):
        self.main, self.pool = main, pool
        self.front = front
        self.segments = []          # [graph, ...]
        self.recording = False
        self.pack_out = (front[0], front[1])
        self._graph = self._ctx = None
        self._n_sample = 0
        self._cut = False

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

    def sample_index_buffer(self, b, m):
        """index buffer of the next _sample call: the group's for the first sampler, None = allocate as usual"""
        if self._n_sample == 0:
            assert tuple(self.front[2].shape) == (b, m)
            return self.front[2]
        return None

    def enter_samplers(self):
        self._n_sample += 1
        self._cut = self._n_sample == 1
        if not self._cut:
            return
        self._graph.capture_end()          # segment 0 ends here; the group launches the sampler between the segments
        self._ctx.__exit__(None, None, None)
        self.recording = True

    def add_sampler(self, xyz, lo, hi, m, scores, gamma, idx_out, idx_offset):
        # launched by the group for all its passes at once: must be the plain input-only D-FPS
        if not (scores is None and lo == 0 and hi == xyz.shape[1] and idx_offset == 0 and m == idx_out.shape[1]
                and xyz.data_ptr() == self.front[1].data_ptr() and idx_out.data_ptr() == self.front[2].data_ptr()):
            raise NotImplementedError("grouped first sampler: expected one d-fps over the whole input cloud")

    def exit_samplers(self):
        if not self._cut:
            return
        self._cut = False
        self.recording = False
        self.segments.append(self._graph)
        self.begin()


class GraphedDet6D(object):
    """One Det6D pass (backbone -> head -> fused post-processing) captured into a hipGraph on its
    own HIP stream: ~130 kernel launches replay with a single host call, so several batches can be
    kept in flight from one Python thread (launch() is asynchronous, finalize() waits).

    The captured graph reads `self.points` (static input, (B*N, 1+3+C)); pass a tensor to launch()
    to have it copied in first, or write into `self.points` yourself."""

    def __init__(self, model, batch_size, n_points, point_width=5, points=None, warmup=2, front=None, stream=None):
        """front = (rows, xyz, idx) slices of a Det6DGroup: the pass packs its points into them and takes the first
        sampler's indices from idx (the group launches that sampler for all its passes)"""
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

        def body():
            bd = {'batch_size': batch_size, 'points': self.points}
            for module in model.module_list:
                bd = module(bd)
            boxes, scores, labels, index, count = fused.postprocess(
                bd['batch_cls_preds'].contiguous(), bd['batch_box_preds'].contiguous(), batch_size, pp.SCORE_THRESH,
                nms.NMS_PRE_MAXSIZE, nms.NMS_POST_MAXSIZE, nms.NMS_THRESH)
            # pred_labels are int64 in the reference (detector3d_template.py:239): converted once, inside the graph
            return bd, (boxes, scores, labels.long(), index, count)

        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(self.stream):
            for _ in range(warmup):
                body()
        self.stream.synchronize()
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
            with torch.no_grad(), torch.cuda.graph(self.graph, stream=self.stream):
                self.batch_dict, (self.boxes, self.scores, self.labels, self.index, self.count) = body()
        self.count_host = torch.empty(self.count.shape, dtype=self.count.dtype, pin_memory=True)
        self.done = torch.cuda.Event()
        self._weights_version = getattr(model, 'weights_version', 0)

    def _relaunched(self):
        lazy = self.batch_dict.get('point_coords_list', None)
        if hasattr(lazy, 'reset'):
            lazy.reset()           # lists built on first access from the centres: stale after a replay

    def _check_weights(self):
        if getattr(self.model, 'weights_version', 0) != self._weights_version:
            raise RuntimeError("the model's weights changed after this pass was captured (load_state_dict / train()): "
                               "its graph still points at the old folded matrices; build a new GraphedDet6D / Det6DGroup")

    def launch_front(self, points=None):
        """segment 0 (everything before the first sampler) on the CURRENT stream (the group's sampler stream), once
        the pass's previous launch has finished with the buffers"""
        self._check_weights()
        self._relaunched()
        torch.cuda.current_stream().wait_event(self.done)
        if callable(points):          # an input producer filling self.points on the current stream (bench.py pipeline leg)
            points(self)
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
            self.done.record()
        return self

    def launch(self, points=None):
        self._check_weights()
        self._relaunched()
        with torch.cuda.stream(self.stream):
            if points is not None and points.data_ptr() != self.points.data_ptr():
                self.points.copy_(points, non_blocking=True)
            if self.segments is not None:
                raise RuntimeError("this pass belongs to a Det6DGroup: launch the group")
            self.graph.replay()
            self.count_host.copy_(self.count, non_blocking=True)
            self.done.record()
        return self

    #: seconds the host spent blocked in finalize() (all passes): host-bound pipelines show ~0 here
    host_wait_s = 0.0

    def finalize(self):
        """pred_dicts of the last launch (views into the graph's static outputs)"""
        if not self.done.query():
            t0 = time.perf_counter()
            self.done.synchronize()
            GraphedDet6D.host_wait_s += time.perf_counter() - t0
        return [{'pred_boxes': self.boxes[i, :k], 'pred_scores': self.scores[i, :k],
                 'pred_labels': self.labels[i, :k]} for i, k in enumerate(self.count_host.tolist())]


class Det6DGroup(object):
    """K captured passes whose FIRST sampler (D-FPS over the input cloud: depends on nothing else) runs as one
    launch over all K x B scenes on a stream of its own; see SAMPLER_GROUP above.  Two stages:
      launch_front(): pack + first sampler of the K passes on `sampler_stream` (waits until the passes' previous
                      launch is done with the buffers);
      launch_rest():  the remaining graph segments of every pass on its main stream, after the sampler.
    Issue launch_front() of later groups BEFORE launch_rest() of earlier ones and the samplers (one 1024-thread
    workgroup per scene, a 3.5 ms latency chain that waits 10-20 ms for a free CU beside GEMM traffic) run ahead of
    the GEMM stage instead of blocking its streams.  launch() = both stages back to back."""

    def __init__(self, model, batch_size, n_points, k, sampler_stream, point_width=5, points=None, main_streams=None):
        from .ops import fused
        from .pcdet.ops.pointnet2.pointnet2_batch.pointnet2_modules import rows_ld
        sa1 = model.backbone_3d.SA_modules[0]
        m1 = sum(sa1.npoint_list)
        ld = rows_ld(point_width - 4)
        dev = 'cuda'
        self.k, self.batch_size, self.n_points, self.m1 = k, batch_size, n_points, m1
        self.hi = sampler_stream
        self.rows_all = torch.empty((k * batch_size, n_points, ld), dtype=torch.float32, device=dev)
        self.xyz_all = torch.empty((k * batch_size, n_points, 3), dtype=torch.float32, device=dev)
        self.idx_all = torch.empty((k * batch_size, m1), dtype=torch.int32, device=dev)
        self.temp_all = fused.fps_workspace(k * batch_size, n_points, dev)
        self.runners = []
        for j in range(k):
            sl = slice(j * batch_size, (j + 1) * batch_size)
            own = points[j % len(points)] if isinstance(points, (list, tuple)) else points   # a static input per pass
            self.runners.append(GraphedDet6D(model, batch_size, n_points, point_width, points=own,
                                             front=(self.rows_all[sl], self.xyz_all[sl], self.idx_all[sl]),
                                             stream=None if main_streams is None else main_streams[j % len(main_streams)]))
        self._fps = fused.fps_fused
        self._sampled = torch.cuda.Event()
        self._count = k

    def launch_front(self, points=None, count=None):
        self._count = self.k if count is None else count
        nb = self._count * self.batch_size
        with torch.cuda.stream(self.hi):
            for r in self.runners[:self._count]:
                r.launch_front(points)
            self._fps(self.xyz_all[:nb], 0, self.n_points, self.m1, None, 1.0, self.idx_all[:nb], 0, temp=self.temp_all)
            self._sampled.record(self.hi)
        return self

    def launch_rest(self):
        active = self.runners[:self._count]
        for r in active:
            r.launch_rest(self._sampled)
        return active

    def launch(self, points=None, count=None):
        return self.launch_front(points, count).launch_rest()


def warn_hw_queues(need):
    """ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable when the
    runtime initialises: with fewer queues than streams the passes in flight serialise"""
    have = int(os.environ.get('GPU_MAX_HW_QUEUES', '4'))
    if have < need:
        import warnings
        warnings.warn("GPU_MAX_HW_QUEUES=%d but the pipeline uses %d HIP streams: they will alias onto the hardware "
                      "queues and serialise; export GPU_MAX_HW_QUEUES=24 before the process touches the GPU" % (have, need),
                      RuntimeWarning, stacklevel=3)
    return have


class ScenePipeline(object):
    """The throughput runner: batches of scenes stream through `n_main / group + prefetch` Det6DGroups.
    Stage 1 of a group (pack + first sampler of its `group` passes) is issued `prefetch` groups ahead on one of the
    sampler streams, stage 2 (the captured rest of every pass) on `n_main` main streams; a pass is finalised (its
    detections sliced per scene, the only host sync) when the slot it occupies is needed again.

    points: None — every pass owns a static input buffer (`passes[i].points`) the caller fills or copies into;
            a tensor — every pass reads it; a list of tensors — pass i reads points[i % len(points)]."""

    def __init__(self, model, batch_size, n_points, n_main=16, group=4, prefetch=4, sampler_streams=6, points=None,
                 point_width=5, main_streams=None, samplers=None):
        self.k = max(1, min(group, n_main))
        self.prefetch = prefetch
        warn_hw_queues(n_main + sampler_streams)
        self.main_streams = list(main_streams) if main_streams else [torch.cuda.Stream() for _ in range(n_main)]
        self.sampler_streams = list(samplers) if samplers else [torch.cuda.Stream() for _ in range(sampler_streams)]
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

    def run(self, steps, feed=None, on_done=None):
        """`steps` passes through the pipeline; `feed` (host tensor or callable(pass)) supplies the input of a pass whose
        static buffer is not resident already; on_done(step, pass, pred_dicts) is called in step order as passes are
        finalised.  Returns the number of passes finalised (== steps)."""
        k, n_groups, prefetch = self.k, self.n_groups, self.prefetch
        counts, left = [], steps
        while left > 0:
            counts.append(min(k, left))
            left -= counts[-1]
        done, inflight = 0, []

        def finish(active):
            nonlocal done
            for r in active:
                preds = r.finalize()
                if on_done is not None:
                    on_done(done, r, preds)
                done += 1

        for g in range(min(prefetch, len(counts))):
            self.groups[g % n_groups].launch_front(feed, counts[g])
        for g in range(len(counts)):
            if len(inflight) >= n_groups - prefetch:
                finish(inflight.pop(0))
            if g + prefetch < len(counts):
                self.groups[(g + prefetch) % n_groups].launch_front(feed, counts[g + prefetch])
            inflight.append(self.groups[g % n_groups].launch_rest())
        for active in inflight:
            finish(active)
        return done
