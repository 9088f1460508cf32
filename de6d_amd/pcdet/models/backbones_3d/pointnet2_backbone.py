"""PointNet2FSMSG backbone (core/pcdet/models/backbones_3d/pointnet2_backbone.py:97-263) on the
fused HIP set-abstraction layers.  Constructor reads the same SA_CONFIG keys; forward reads and
writes the same batch_dict keys.  Differences by design: the flat `points`
tensor is packed once into point rows by a HIP kernel (no per-scene `.sum()` host syncs as in
:216-219; scenes must hold equal point counts, which the reference asserts at :219), and the
rows of the last layer travel to the head under `batch_dict['_det6d_rows']`."""
import torch
import torch.nn as nn

from ...ops.pointnet2.pointnet2_batch import pointnet2_modules
from ...ops_backend import fused


class _LazyCoords(list):
    """point_coords_list: (B*M, 4) [batch index, x, y, z] tensors of the SA levels, built from the (B, M, 3) centres on
    first access"""

    def __init__(self, xyz_levels):
        super().__init__([None] * len(xyz_levels))
        self._xyz = list(xyz_levels)

    def reset(self):
        """a captured pass was replayed: the centres changed under the cached lists (runtime.GraphedDet6D)"""
        for i in range(len(self)):
            list.__setitem__(self, i, None)

    def _get(self, i):
        v = list.__getitem__(self, i)
        if v is None:
            v = fused.with_batch_index(self._xyz[i], 3)
            list.__setitem__(self, i, v)
        return v

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._get(j) for j in range(*i.indices(len(self)))]
        return self._get(i if i >= 0 else len(self) + i)

    def __iter__(self):
        return (self._get(i) for i in range(len(self)))


class PointNet2FSMSG(nn.Module):
    def __init__(self, model_cfg, input_channels, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        sa = model_cfg.SA_CONFIG
        use_xyz = sa.get('USE_XYZ', True)
        dilated_group = sa.get('DILATED_RADIUS_GROUP', False)
        skip_connection = sa.get('SKIP_CONNECTION', False)
        weight_gamma = sa.get('WEIGHT_GAMMA', 1.0)
        agg_cfg = sa.get('AGGREGATION_MLPS', None)
        conf_cfg = sa.get('CONFIDENCE_MLPS', None)

        self.SA_modules = nn.ModuleList()
        self.num_points_each_layer = []
        channel_in = input_channels - 3
        self.input_feature_channels = channel_in
        skip_channels = [channel_in]
        for k in range(len(sa.NPOINT_LIST)):
            mlps = [[channel_in] + list(spec) for spec in sa.MLPS[k]]
            channel_out = sum(spec[-1] for spec in mlps) + (channel_in if skip_connection else 0)
            aggregation_mlp = list(agg_cfg[k]) if agg_cfg and agg_cfg[k] else None
            if aggregation_mlp:
                channel_out = aggregation_mlp[-1]
            confidence_mlp = list(conf_cfg[k]) if conf_cfg and conf_cfg[k] else None
            self.SA_modules.append(pointnet2_modules.PointnetSAModuleFSMSG(
                npoint_list=sa.NPOINT_LIST[k], sample_range_list=sa.SAMPLE_RANGE_LIST[k],
                sample_method_list=sa.SAMPLE_METHOD_LIST[k], radii=sa.RADIUS[k], nsamples=sa.NSAMPLE[k],
                mlps=mlps, use_xyz=use_xyz, dilated_radius_group=dilated_group,
                skip_connection=skip_connection, weight_gamma=weight_gamma,
                aggregation_mlp=aggregation_mlp, confidence_mlp=confidence_mlp))
            self.num_points_each_layer.append(sum(sa.NPOINT_LIST[k]))
            skip_channels.append(channel_out)
            channel_in = channel_out
        self.num_point_features = channel_out

        fp_mlps = model_cfg.get('FP_MLPS', None)
        self.FP_modules = None
        if fp_mlps is not None:
            self.FP_modules = nn.ModuleList()
            l_skipped = len(sa.NPOINT_LIST) - len(fp_mlps)
            for k in range(len(fp_mlps)):
                pre_channel = fp_mlps[k + 1][-1] if k + 1 < len(fp_mlps) else channel_out
                self.FP_modules.append(pointnet2_modules.PointnetFPModule(
                    mlp=[pre_channel + skip_channels[k + l_skipped]] + list(fp_mlps[k])))
            self.num_point_features = fp_mlps[0][-1]

    def break_up_pc(self, pc):
        batch_idx = pc[:, 0]
        xyz = pc[:, 1:4].contiguous()
        features = pc[:, 4:].contiguous() if pc.size(-1) > 4 else None
        return batch_idx, xyz, features

    @staticmethod
    def _with_batch_column(x):
        """(B,M,3) -> (B*M,4) rows [batch_idx, x, y, z]"""
        return fused.with_batch_index(x, 3)

    def forward(self, batch_dict):
        batch_size = batch_dict['batch_size']
        points = batch_dict['points']
        assert points.shape[0] % batch_size == 0, 'every scene must hold the same number of points'
        n = points.shape[0] // batch_size
        c_in = points.shape[1] - 4
        ld = pointnet2_modules.rows_ld(c_in)
        rows, xyz = fused.pack_points(points.contiguous(), ld)
        rows = rows.view(batch_size, n, ld)
        xyz = xyz.view(batch_size, n, 3)

        l_xyz, l_rows, l_scores = [xyz], [rows], [None]
        for sa in self.SA_modules:
            nx, nr, ns = sa.forward_rows(l_xyz[-1], l_rows[-1], scores=l_scores[-1])
            l_xyz.append(nx)
            l_rows.append(nr)
            l_scores.append(ns)

        # [batch index, x, y, z] rows of every level: built on first access (a list subclass), the inference path reads
        # none of them and every small launch costs ~4 us of a step with many passes in flight
        batch_dict['point_coords_list'] = _LazyCoords(l_xyz[1:])
        batch_dict['point_scores_list'] = [None if s is None else s.reshape(-1, 1) for s in l_scores[1:]]

        if self.FP_modules is not None:
            feats = [None if c_in == 0 else l_rows[0][:, :, 3:3 + c_in].transpose(1, 2).contiguous()]
            chans = [c_in] + [m._folded['out_channels'] for m in self.SA_modules]
            for r, c in zip(l_rows[1:], chans[1:]):
                feats.append(r[:, :, 3:3 + c].transpose(1, 2).contiguous())
            for i in range(-1, -(len(self.FP_modules) + 1), -1):
                feats[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], feats[i - 1], feats[i])
            out_idx = -(len(self.FP_modules) + 1)
            point_features = feats[out_idx].permute(0, 2, 1).contiguous()
            out_xyz = l_xyz[out_idx]
        else:
            c_out = self.num_point_features
            last = l_rows[-1]
            point_features = last.view(-1, last.shape[-1])[:, 3:3 + c_out]   # (B*M, C) strided view of the rows, no copy
            out_xyz = l_xyz[-1]
            batch_dict['_det6d_rows'] = (out_xyz, l_rows[-1])
        batch_dict['point_features'] = point_features.view(-1, point_features.shape[-1]) if point_features.is_contiguous() else point_features
        batch_dict['point_coords'] = self._with_batch_column(out_xyz)
        batch_dict['point_scores'] = l_scores[-1]
        return batch_dict
