"""Set-abstraction / feature-propagation modules of
core/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py (PointnetSAModuleFSMSG :497-607 with
the forward of _PointnetSAModuleFSBase :358-494, PointnetFPModule :124-174), re-designed around
the fused HIP ops.

The nn.Conv/BatchNorm children exist only to own parameters under the reference's state-dict
names (`mlps.0.0.weight`, `aggregation_mlp.1.running_var`, ...): they are never called.  At first
use the BN statistics are folded into (K, N) weight matrices + shift vectors that live on the
device, and the whole layer runs as: FPS -> row gather -> ball query -> [gather + 3x(GEMM, shift,
ReLU) + mask + max-pool] per radius group -> aggregation GEMM -> confidence GEMMs.

Internal layout ("rows"): one row per point, `[x, y, z, f_0 .. f_{C-1}, 0-pad]`, row stride a
multiple of 4 floats.  A neighbour gather is then one contiguous row read, and `xyz - centre`
touches only the first three columns.  `forward()` keeps the reference's channel-major
signature by converting at the boundary; `forward_rows()` is the fast path the backbone uses.
"""
from typing import List

import numpy as np
import os

import torch
import torch.nn as nn

from ....ops_backend import fused, pointnet2_batch_hip as pn2
from . import pointnet2_utils


def round4(v):
    return (v + 3) // 4 * 4


def rows_ld(channels):
    """row stride of a rows tensor carrying `channels` feature channels"""
    return round4(3 + channels)


def _np(t):
    return t.detach().cpu().numpy().astype(np.float32)


def fold_layer(conv, bn, k_rows, k_offset=0):
    """Fold Conv(k=1) [+ BatchNorm(eval)] into W (k_rows, round4(Cout)) and shift (Cout,), fp32.

    y = BN(conv(x)) = x @ (W_conv * s)^T + (beta - mean * s),  s = gamma / sqrt(var + eps);
    rows [k_offset, k_offset + Cin) of W hold the folded weights, every other row is zero
    (xyz columns a layer does not consume, and the row padding).
    """
    w = _np(conv.weight).reshape(conv.weight.shape[0], -1)  # (Cout, Cin)
    cout, cin = w.shape
    if bn is not None:
        scale = _np(bn.weight) / np.sqrt(_np(bn.running_var) + np.float32(bn.eps))
        shift = _np(bn.bias) - _np(bn.running_mean) * scale
        w = w * scale[:, None]
        if conv.bias is not None:
            shift = shift + _np(conv.bias) * scale
    else:
        shift = _np(conv.bias) if conv.bias is not None else np.zeros(cout, np.float32)
    mat = np.zeros((k_rows, round4(cout)), np.float32)
    mat[k_offset:k_offset + cin, :cout] = w.T
    return mat, shift.astype(np.float32), cout


def fold_sequential(seq, k_rows, k_offset=0):
    """[(W, shift, cout, act)] for a Sequential of Conv(/BN/ReLU) blocks; later layers take the
    previous layer's padded width as their K."""
    layers, mods, i = [], list(seq), 0
    while i < len(mods):
        conv = mods[i]
        assert isinstance(conv, (nn.Conv1d, nn.Conv2d))
        bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], (nn.BatchNorm1d, nn.BatchNorm2d)) else None
        j = i + (2 if bn is not None else 1)
        act = 1 if j < len(mods) and isinstance(mods[j], nn.ReLU) else 0
        if act:
            j += 1
        mat, shift, cout = fold_layer(conv, bn, k_rows, k_offset)
        layers.append((mat, shift, cout, act))
        k_rows, k_offset = round4(cout), 0
        i = j
    return layers


def to_device(layers, device):
    return [(torch.from_numpy(m).to(device), torch.from_numpy(s).to(device), c, a) for m, s, c, a in layers]


def run_chain(x, layers, out=None, col0=0):
    """plain GEMM chain over rows; the last layer may write into `out` at column `col0`"""
    for li, (w, shift, cout, act) in enumerate(layers):
        last = li == len(layers) - 1
        if last and out is not None:
            fused.linear(x, w, shift, act, out, ncols=cout, col0=col0)
            return out
        y = torch.empty((x.numel() // x.shape[-1], w.shape[1]), dtype=torch.float32, device=x.device)
        # padded columns feed the next layer's zero weight rows: the kernel writes them as zeros
        fused.linear(x, w, shift, act, y, ncols=cout, ncols_pad=w.shape[1] if w.shape[1] != cout else 0)
        x = y
    return x


#: The samplers of a layer run one after the other on the caller's stream.  Forking them onto side streams
#: (DET6D_FORKED_SAMPLERS=1) shortens one pass by ~0.45 ms but costs throughput with many passes in flight
#: (4121 vs 4300 scenes/s at 15 passes: more sampler workgroups resident at once, fork/join in every graph).
SEQUENTIAL_SAMPLERS = fused.L.experiment_switch('DET6D_FORKED_SAMPLERS') is None

#: Grouped MLPs run on compact (ragged) row lists: a ball with cnt < nsample hits is padded by the reference with
#: repetitions of its first cnt hits, so only the first 2^ceil(log2 cnt) slots of a centre are evaluated — the
#: pooled features are identical bit for bit (csrc/compact.hip).  DET6D_DENSE_ROWS=1 restores the reference's
#: dense (B, m, nsample) row space.
COMPACT_ROWS = os.environ.get('DET6D_DENSE_ROWS') is None

#: First layer of a grouped MLP from per-point partial sums (csrc/expand.hip): one plain GEMM over the N points of the
#: layer + 3 FMAs per grouped output instead of a (3 + C)-deep GEMM over every (centre, neighbour) row; identical bits
#: (the chain order of gathered rows puts the relative coordinates last).  DET6D_NO_EXPAND=1: the gathered GEMM instead.
EXPAND_FIRST_LAYER = fused.L.experiment_switch('DET6D_NO_EXPAND') is None


class _PointnetSAModuleFSBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.groupers = None
        self.mlps = None
        self.npoint_list = []
        self.sample_range_list = [[0, -1]]
        self.sample_method_list = ['d-fps']
        self.radii = []
        self.pool_method = 'max_pool'
        self.dilated_radius_group = False
        self.weight_gamma = 1.0
        self.skip_connection = False
        self.aggregation_mlp = None
        self.confidence_mlp = None
        self._folded = None
        self._side_streams = None

    # ---- weight preparation -------------------------------------------------------------
    def invalidate(self):
        self._folded = None

    def _prepare(self, device):
        if self._folded is not None and self._folded['device'] == device:
            return self._folded
        if self.training:
            raise RuntimeError("the HIP set-abstraction path folds BatchNorm: call .eval() first")
        in_ld = rows_ld(self.in_channels)
        groups = [to_device(fold_sequential(seq, in_ld), device) for seq in self.mlps]
        pooled_width = sum(g[-1][2] for g in groups)
        out_channels = pooled_width
        agg = conf = None
        if self.aggregation_mlp is not None:
            agg = to_device(fold_sequential(self.aggregation_mlp, round4(pooled_width)), device)
            out_channels = agg[-1][2]
        if self.confidence_mlp is not None:
            conf = to_device(fold_sequential(self.confidence_mlp, rows_ld(out_channels), k_offset=3), device)
        # groups whose first layer runs as "per-point GEMM + expand" (not the ones a fused chain kernel takes whole)
        expand, pcols, col = [], {}, 0
        for gi, (layers, ns) in enumerate(zip(groups, self.nsamples)):
            if COMPACT_ROWS and ns in (4, 8, 16, 32):
                chained = fused.chain_compact_eligible(in_ld, layers)
            else:
                chained = fused.chain_eligible(in_ld, layers, ns)
            if EXPAND_FIRST_LAYER and not chained and len(layers) >= 2 and layers[0][2] % 4 == 0 and in_ld > 4:
                expand.append(gi)
                pcols[gi] = col
                col += layers[0][0].shape[1]
        p_w = None
        if expand:
            p_w = torch.cat([groups[gi][0][0] for gi in expand], dim=1).clone()
            p_w[:3] = 0           # the coordinate rows enter in the expand step; P is the chain over the feature columns
        self._folded = dict(device=device, groups=groups, pooled_width=pooled_width, agg=agg, conf=conf,
                            out_channels=out_channels, expand=expand, pcols=pcols, p_w=p_w)
        return self._folded

    # ---- sampling -----------------------------------------------------------------------
    def _sample_one(self, xyz, scores, lo, hi, method, npoint, idx_out, offset):
        """one sampler -> idx_out[:, offset:offset+npoint]; slice, sigmoid**gamma weights, 1e10 init and
        the + lo offset (pointnet2_modules.py:380,415-424,448) all happen inside det6d_fps_fused"""
        hi = xyz.shape[1] if hi == -1 else hi
        if method == 'd-fps':
            fused.fps_fused(xyz, lo, hi, npoint, None, 1.0, idx_out, offset)
        elif method == 's-fps':
            assert scores is not None
            fused.fps_fused(xyz, lo, hi, npoint, scores, self.weight_gamma, idx_out, offset)
        else:
            raise NotImplementedError(
                "sampling method %r is outside the Det6D hot path (SURVEY.md 2.1 #8)" % method)

    def _sample(self, xyz, scores):
        """fusion sampling (pointnet2_modules.py:376-450).  The samplers of one layer are independent
        latency chains on one workgroup per scene, so they run concurrently on forked HIP streams
        (also under hipGraph capture, where the fork/join becomes two parallel branches)."""
        jobs = list(zip(self.sample_range_list, self.sample_method_list, self.npoint_list))
        b = xyz.shape[0]
        ctl = fused.SAMPLER_SEGMENTS                                  # capture controller of a pass (runtime.py), or None
        # the controller counts EVERY layer (its index buffers and hoisted samplers are keyed by layer number) ...
        layer = ctl.next_layer() if ctl is not None else -1
        if ctl is not None and not (len(jobs) == 1 or SEQUENTIAL_SAMPLERS):
            # ... but it cannot serve a layer whose samplers run on forked streams (DET6D_FORKED_SAMPLERS=1): its hoisted
            # picks would be recomputed into a private buffer and the group's buffers ignored
            raise RuntimeError("DET6D_FORKED_SAMPLERS=1 cannot be combined with captured passes (GraphedDet6D / Det6DGroup "
                               "hoist the input-only samplers): unset it, or run the model eagerly")
        idx = ctl.index_buffer(layer, b, sum(self.npoint_list)) if ctl is not None else None
        if idx is not None:
            assert tuple(idx.shape) == (b, sum(self.npoint_list)), "index buffer of layer %d has the wrong shape" % layer
        if idx is None:
            idx = torch.empty((b, sum(self.npoint_list)), dtype=torch.int32, device=xyz.device)
        offsets = [sum(self.npoint_list[:i]) for i in range(len(jobs))]
        if len(jobs) == 1 or SEQUENTIAL_SAMPLERS:
            for j, (((lo, hi), method, npoint), off) in enumerate(zip(jobs, offsets)):
                if ctl is not None and ctl.hoisted(layer, j):
                    continue          # launched by the group for all its passes, ahead of this segment (runtime.hoist_plan)
                self._sample_one(xyz, scores, lo, hi, method, npoint, idx, off)
            if ctl is not None:
                ctl.after_samplers(layer, xyz, idx)    # single-graph passes: fork / join of the input-only sampler chain
            return idx
        main = torch.cuda.current_stream()
        if self._side_streams is None or len(self._side_streams) < len(jobs) - 1:
            self._side_streams = [torch.cuda.Stream() for _ in range(len(jobs) - 1)]
        for i, ((lo, hi), method, npoint) in enumerate(jobs[1:], start=1):
            side = self._side_streams[i - 1]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._sample_one(xyz, scores, lo, hi, method, npoint, idx, offsets[i])
        (lo, hi), method, npoint = jobs[0]
        self._sample_one(xyz, scores, lo, hi, method, npoint, idx, 0)
        for side in self._side_streams[:len(jobs) - 1]:
            main.wait_stream(side)
        return idx

    # ---- fast path ----------------------------------------------------------------------
    def forward_rows(self, xyz, rows, scores=None, new_xyz=None):
        """
        xyz (B,N,3), rows (B,N,ld) [xyz | features | pad]  ->
        new_xyz (B,M,3), new_rows (B,M,ld') [xyz | new features | pad] or pooled (B,M,sumC) when
        there is no aggregation MLP, new_scores (B,M) or None
        """
        if self.pool_method != 'max_pool' or self.skip_connection:
            raise NotImplementedError("only max_pool without skip connection is on the Det6D path")
        f = self._prepare(rows.device)
        b, n, _ = xyz.shape
        new_rows = None
        if new_xyz is None:
            sample_idx = self._sample(xyz, scores)
            m = sample_idx.shape[1]
            if f['agg'] is not None:  # next level's rows: xyz now, features by the aggregation GEMM, pad zeroed
                ld_next = rows_ld(f['out_channels'])
                new_rows = torch.empty((b, m, ld_next), dtype=torch.float32, device=rows.device)
                new_xyz = fused.gather_centres(xyz, sample_idx, new_rows, 3 + f['out_channels'])
            else:
                new_xyz = fused.gather_centres(xyz, sample_idx)
        m = new_xyz.shape[1]
        pooled = torch.empty((b * m, round4(f['pooled_width'])), dtype=torch.float32, device=rows.device)
        if pooled.shape[1] != f['pooled_width']:
            pooled[:, f['pooled_width']:].zero_()
        col = 0
        # neighbour search: shells [former, radius) when dilated, plain balls otherwise
        shells, former_radius = [], 0.0
        for radius, nsample in zip(self.radii, self.nsamples):
            shells.append((former_radius if self.dilated_radius_group else 0.0, radius, nsample))
            former_radius = radius
        counted = None
        widths = [layers[-1][2] for layers in f['groups']]
        if len(shells) == 2:
            fast = None
            if COMPACT_ROWS and (widths[0] | widths[1] | pooled.shape[1]) % 4 == 0:
                # compact-row engine on a large cloud: the query counts the list builder's parts and skips the padding slots
                fast = fused.ball_query_pair_lists(xyz, new_xyz, shells[0], shells[1])
            if fast is not None:
                ca, ia, cb, ib = fast[:4]
                counted = fast[4:]
            else:
                ca, ia, cb, ib = fused.ball_query_pair(xyz, new_xyz, shells[0], shells[1])
            found = [(ca, ia), (cb, ib)]
        else:
            found = []
            for rin, rout, nsample in shells:
                idx_cnt = torch.zeros((b, m), dtype=torch.int32, device=xyz.device)
                idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=xyz.device)
                if self.dilated_radius_group:
                    pn2.ball_query_dilated_wrapper(b, n, m, rin, rout, nsample, new_xyz, xyz, idx_cnt, idx)
                else:
                    pn2.ball_query_cnt_wrapper(b, n, m, rout, nsample, new_xyz, xyz, idx_cnt, idx)
                found.append((idx_cnt, idx))
        lists = [None] * len(found)
        if (COMPACT_ROWS and fused.COMPACT_SPLIT and len(found) == 2 and all(ns in (4, 8, 16, 32) for ns in self.nsamples)
                and (widths[0] | widths[1] | pooled.shape[1]) % 4 == 0):
            # both groups' lists in one pair of launches; their slices of `pooled` are cleared by the builder
            lists = fused.compact_groups_pair(found, n, pooled, [(0, widths[0]), (widths[0], widths[1])], counted=counted)
        p_all = None
        if f['expand']:   # per-point partial sums of the first layers of all expand groups: one plain GEMM over the points
            p_all = torch.empty((b * n, f['p_w'].shape[1]), dtype=torch.float32, device=rows.device)
            fused.linear(rows.view(b * n, rows.shape[-1]), f['p_w'], None, 0, p_all)
        for gi, ((idx_cnt, idx), nsample, layers, cr) in enumerate(zip(found, self.nsamples, f['groups'], lists)):
            if COMPACT_ROWS and nsample in (4, 8, 16, 32):
                # parts of a centre are combined by an atomic max: the group's slice of `pooled` is cleared by the list builder
                w_out = layers[-1][2]
                if cr is not None:
                    pass
                elif fused.COMPACT_SPLIT and (col | w_out | pooled.shape[1]) % 4 == 0:
                    cr = fused.compact_groups(idx_cnt, idx, n, zero=(pooled, col, w_out))
                else:
                    if fused.COMPACT_SPLIT:
                        pooled[:, col:col + w_out].zero_()
                    cr = fused.compact_groups(idx_cnt, idx, n)
                if fused.chain_compact_eligible(rows.shape[-1], layers):
                    fused.mlp_chain3_compact(rows, cr, new_xyz, layers, pooled, col)
                    col += layers[-1][2]
                    continue
                if gi in f['expand'] and fused.group_kernel_eligible(layers, nsample, True):
                    fused.mlp_group3(p_all, f['pcols'][gi], layers, rows, new_xyz, pooled, col, compact=cr)
                    col += layers[-1][2]
                    continue
                x = None
                for li, (w, shift, cout, act) in enumerate(layers):
                    if li == len(layers) - 1:
                        tgt, kw = pooled, dict(ncols=cout, col0=col, cnt=idx_cnt, pool=-1)
                    else:
                        tgt = torch.empty((cr.capacity, w.shape[1]), dtype=torch.float32, device=rows.device)
                        kw = dict(ncols=cout, ncols_pad=w.shape[1] if w.shape[1] != cout else 0)
                    if li == 0 and gi in f['expand']:
                        fused.group_expand(p_all, f['pcols'][gi], w, shift, act, cout, rows, new_xyz, tgt, compact=cr)
                    elif li == 0:
                        fused.linear(rows, w, shift, act, tgt, ctr=new_xyz, compact=cr, gather=True, **kw)
                    else:
                        fused.linear(x, w, shift, act, tgt, compact=cr, **kw)
                    x = tgt
                col += layers[-1][2]
                continue
            if fused.chain_eligible(rows.shape[-1], layers, nsample):   # narrow group: one fused launch
                fused.mlp_chain3(rows, idx, new_xyz, idx_cnt, layers, pooled, col)
                col += layers[-1][2]
                continue
            if gi in f['expand'] and fused.group_kernel_eligible(layers, nsample, False) and (nsample == 32 or m % 2 == 0):
                fused.mlp_group3(p_all, f['pcols'][gi], layers, rows, new_xyz, pooled, col, idx=idx, cnt=idx_cnt)
                col += layers[-1][2]
                continue
            x = None
            for li, (w, shift, cout, act) in enumerate(layers):
                last = li == len(layers) - 1
                poolable = nsample in (8, 16, 32)
                if last and poolable:
                    tgt, kw = pooled, dict(ncols=cout, col0=col, cnt=idx_cnt, pool=nsample)
                else:
                    tgt = torch.empty((b * m * nsample, w.shape[1]), dtype=torch.float32, device=rows.device)
                    kw = dict(ncols=cout, ncols_pad=w.shape[1] if w.shape[1] != cout else 0)
                if li == 0 and gi in f['expand'] and not last:
                    fused.group_expand(p_all, f['pcols'][gi], w, shift, act, cout, rows, new_xyz, tgt, idx=idx)
                elif li == 0:
                    fused.linear(rows, w, shift, act, tgt, idx=idx, ctr=new_xyz, **kw)
                else:
                    fused.linear(x, w, shift, act, tgt, **kw)
                x = tgt
                if last and not poolable:  # any other nsample: the layer is written out, mask + max in their own launch
                    fused.group_maxpool(x, nsample, cout, idx_cnt, pooled, col)
            col += layers[-1][2]
        new_scores = None
        if f['agg'] is not None:
            if new_rows is None:  # centres supplied by the caller
                new_rows = torch.zeros((b, m, rows_ld(f['out_channels'])), dtype=torch.float32, device=rows.device)
                new_rows[:, :, :3] = new_xyz
            if f['conf'] is not None and f['conf'][-1][2] == 1 and len(f['agg']) == 1:
                # aggregation + confidence chain in ONE launch (csrc/mlp_rows.hip): the aggregated features go to the next
                # level's rows AND stay in LDS as the input of the confidence layers (whose first three weight rows, the
                # coordinates', are zero: the chain starts at weight row 3)
                wa, sha, ca, aa = f['agg'][0]
                new_scores = torch.empty((b * m, 1), dtype=torch.float32, device=rows.device)
                spec = [(wa, 0, sha, f['pooled_width'], ca, aa, new_rows.view(b * m, -1), 3)]
                kin, wrow0 = ca, 3
                for li, (w, sh, cout, act) in enumerate(f['conf']):
                    spec.append((w, wrow0, sh, kin, cout, act, new_scores if li == len(f['conf']) - 1 else None, 0))
                    kin, wrow0 = cout, 0
                if fused.mlp_rows_eligible(f['pooled_width'], [spec]):
                    fused.mlp_rows(pooled, 0, [spec])
                    return new_xyz, new_rows, new_scores.view(b, m)
                new_scores = None
            run_chain(pooled, f['agg'], out=new_rows, col0=3)
            if f['conf'] is not None and f['conf'][-1][2] == 1:   # the last layer writes the (B*M, 1) score column itself
                new_scores = torch.empty((b * m, 1), dtype=torch.float32, device=rows.device)
                run_chain(new_rows, f['conf'], out=new_scores)
                new_scores = new_scores.view(b, m)
            elif f['conf'] is not None:
                new_scores = run_chain(new_rows, f['conf'])[:, 0].reshape(b, m).contiguous()
            return new_xyz, new_rows, new_scores
        return new_xyz, pooled.view(b, m, -1), None

    # ---- reference-shaped signature -----------------------------------------------------
    def forward(self, xyz, features=None, new_xyz=None, scores=None):
        """(B,N,3), (B,C,N) -> new_xyz (B,M,3), new_features (B,C',M), new_scores (B,M) | None"""
        b, n, _ = xyz.shape
        c = 0 if features is None else features.shape[1]
        assert c == self.in_channels, "feature channels %d != %d" % (c, self.in_channels)
        rows = torch.zeros((b, n, rows_ld(c)), dtype=torch.float32, device=xyz.device)
        rows[:, :, :3] = xyz
        if c:
            rows[:, :, 3:3 + c] = features.transpose(1, 2)
        f = self._prepare(rows.device)
        new_xyz, out, new_scores = self.forward_rows(xyz.contiguous(), rows, scores, new_xyz)
        if f['agg'] is not None:
            new_features = out[:, :, 3:3 + f['out_channels']]
        else:
            new_features = out[:, :, :f['pooled_width']]
        return new_xyz, new_features.transpose(1, 2).contiguous(), new_scores


class PointnetSAModuleFSMSG(_PointnetSAModuleFSBase):
    """Set abstraction with fusion sampling and multi-scale grouping (keyword signature of the
    reference, pointnet2_modules.py:500-514)."""

    def __init__(self, *, npoint_list: List[int] = None, sample_range_list: List[List[int]] = None,
                 sample_method_list: List[str] = None, radii: List[float], nsamples: List[int],
                 mlps: List[List[int]], bn: bool = True, use_xyz: bool = True, pool_method='max_pool',
                 dilated_radius_group: bool = False, skip_connection: bool = False, weight_gamma: float = 1.0,
                 aggregation_mlp: List[int] = None, confidence_mlp: List[int] = None):
        super().__init__()
        assert npoint_list is None or len(npoint_list) == len(sample_range_list) == len(sample_method_list)
        assert len(radii) == len(nsamples) == len(mlps)
        if not use_xyz:
            raise NotImplementedError("use_xyz=False is not on the Det6D path")
        self.npoint_list = npoint_list
        self.sample_range_list = sample_range_list
        self.sample_method_list = sample_method_list
        self.radii = list(radii)
        self.nsamples = list(nsamples)
        self.pool_method = pool_method
        self.dilated_radius_group = dilated_radius_group
        self.skip_connection = skip_connection
        self.weight_gamma = weight_gamma
        self.in_channels = mlps[0][0]

        self.groupers = nn.ModuleList()  # parameter-free; kept for structural parity
        self.mlps = nn.ModuleList()
        former_radius, out_channels = 0.0, 0
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            if dilated_radius_group:
                self.groupers.append(pointnet2_utils.QueryAndGroupDilated(former_radius, radius, nsample, use_xyz=True))
            else:
                self.groupers.append(pointnet2_utils.QueryWithCntAndGroup(radius, nsample, use_xyz=True))
            former_radius = radius
            widths = [spec[0] + 3] + list(spec[1:])
            block = []
            for cin, cout in zip(widths[:-1], widths[1:]):
                block += [nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU()]
            self.mlps.append(nn.Sequential(*block))
            out_channels += widths[-1]
        if skip_connection:
            out_channels += self.in_channels

        def conv1d_stack(cin, widths):
            block = []
            for cout in widths:
                block += [nn.Conv1d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm1d(cout), nn.ReLU()]
                cin = cout
            return block, cin

        if aggregation_mlp is not None:
            block, out_channels = conv1d_stack(out_channels, aggregation_mlp)
            self.aggregation_mlp = nn.Sequential(*block)
        if confidence_mlp is not None:
            block, last = conv1d_stack(out_channels, confidence_mlp)
            block.append(nn.Conv1d(last, 1, kernel_size=1, bias=True))
            self.confidence_mlp = nn.Sequential(*block)


class PointnetFPModule(nn.Module):
    """Feature propagation: three_nn + inverse-distance three_interpolate + shared MLP
    (pointnet2_modules.py:124-174)."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        block = []
        for cin, cout in zip(mlp[:-1], mlp[1:]):
            block += [nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU()]
        self.mlp = nn.Sequential(*block)
        self._folded = None

    def invalidate(self):
        self._folded = None

    def forward(self, unknown, known, unknow_feats, known_feats):
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n) | None, known_feats (B,C2,m) -> (B,C',n)"""
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight.contiguous())
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        feats = interpolated if unknow_feats is None else torch.cat([interpolated, unknow_feats], dim=1)
        b, c, n = feats.shape
        if self.training:
            raise RuntimeError("the HIP feature-propagation path folds BatchNorm: call .eval() first")
        if self._folded is None or self._folded[0] != feats.device:
            self._folded = (feats.device, to_device(fold_sequential(self.mlp, round4(c)), feats.device))
        x = torch.zeros((b * n, round4(c)), dtype=torch.float32, device=feats.device)
        x[:, :c] = feats.transpose(1, 2).reshape(b * n, c)
        y = run_chain(x, self._folded[1])
        cout = self._folded[1][-1][2]
        return y[:, :cout].reshape(b, n, cout).transpose(1, 2).contiguous()
