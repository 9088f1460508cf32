"""PointHeadBox6DVote — the ground-aware vote-based 6-DoF point head
(core/pcdet/models/dense_heads/point_head_box6d_vote.py:14-99 constructor, :778-903 forward),
inference branch, on fused HIP ops:

  candidates = first SAMPLE_RANGE points -> vote FC (GEMM) -> clamp + add (kernel) -> SA layer
  around the votes (ball query + fused grouped GEMMs + max-pool) -> shared / cls / reg FCs (GEMMs)
  -> PointBinResidual6DCoder.decode (kernel).

Target assignment and losses (:101-776) are training-only and out of scope.
Parameters live under the reference's names (vote_layers, SA_module.mlps, shared_fc_layer,
cls_layers, reg_layers) so reference checkpoints load unchanged."""
import torch
import torch.nn as nn

from ...ops.pointnet2.pointnet2_batch import pointnet2_modules
from ...ops.pointnet2.pointnet2_batch.pointnet2_modules import fold_sequential, rows_ld, round4, run_chain, to_device
from ...ops_backend import fused
from ...utils import box_coder_utils


class PointHeadBox6DVote(nn.Module):
    def __init__(self, num_class, input_channels, model_cfg, predict_boxes_when_training=False, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.input_channels = input_channels
        self.predict_boxes_when_training = predict_boxes_when_training
        target_cfg = model_cfg.TARGET_CONFIG
        self.box_coder = getattr(box_coder_utils, target_cfg.BOX_CODER)(**target_cfg.BOX_CODER_CONFIG)

        self.vote_cfg = model_cfg.VOTE_CONFIG
        self.vote_layers = self.make_fc_layers(input_channels, 3, self.vote_cfg.VOTE_FC)

        self.sa_cfg = model_cfg.SA_CONFIG
        mlps = [[input_channels] + list(spec) for spec in self.sa_cfg.MLPS]
        self.SA_module = pointnet2_modules.PointnetSAModuleFSMSG(
            radii=self.sa_cfg.RADIUS, nsamples=self.sa_cfg.NSAMPLE, mlps=mlps, use_xyz=True,
            bn=model_cfg.USE_BN)
        channel_in = sum(spec[-1] for spec in mlps)

        shared = []
        for width in model_cfg.SHARED_FC:
            shared += [nn.Conv1d(channel_in, width, kernel_size=1, bias=False), nn.BatchNorm1d(width), nn.ReLU()]
            channel_in = width
        self.shared_fc_layer = nn.Sequential(*shared)
        loss_cls = model_cfg.get('LOSS_CONFIG', {}).get('LOSS_CLS', None)
        cls_out = num_class + 1 if loss_cls == 'CrossEntropy' else num_class
        self.cls_layers = self.make_fc_layers(channel_in, cls_out, model_cfg.CLS_FC)
        self.reg_layers = self.make_fc_layers(channel_in, self.box_coder.code_size, model_cfg.REG_FC)
        self.init_weights()
        self.forward_ret_dict = None
        self._folded = None

    @staticmethod
    def make_fc_layers(input_channels, output_channels, fc_list):
        layers, pre = [], input_channels
        for width in fc_list:
            layers += [nn.Conv1d(pre, width, kernel_size=1, bias=False), nn.BatchNorm1d(width), nn.ReLU()]
            pre = width
        layers.append(nn.Conv1d(pre, output_channels, kernel_size=1, bias=True))
        return nn.Sequential(*layers)

    def init_weights(self):
        # xavier-normal like the reference (point_head_box6d_vote.py:80-99)
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def invalidate(self):
        self._folded = None

    def _prepare(self, device):
        if self._folded is not None and self._folded['device'] == device:
            return self._folded
        if self.training:
            raise RuntimeError("the HIP head folds BatchNorm: call .eval() first")
        ld = rows_ld(self.input_channels)
        sa_width = sum(seq[-3].out_channels for seq in self.SA_module.mlps)
        shared = to_device(fold_sequential(self.shared_fc_layer, round4(sa_width)), device)
        k = round4(shared[-1][2])
        self._folded = dict(
            device=device,
            vote=to_device(fold_sequential(self.vote_layers, ld, k_offset=3), device),
            shared=shared,
            cls=to_device(fold_sequential(self.cls_layers, k), device),
            reg=to_device(fold_sequential(self.reg_layers, k), device))
        return self._folded

    def forward(self, batch_dict):
        if self.training:
            raise NotImplementedError("PointHeadBox6DVote on HIP implements the inference branch only")
        batch_size = batch_dict['batch_size']
        stash = batch_dict.get('_det6d_rows', None)
        if stash is not None:
            xyz, rows = stash
        else:  # rebuild the rows layout from the public keys
            coords = batch_dict['point_coords']
            feats = batch_dict['point_features']
            n = coords.shape[0] // batch_size
            xyz = coords[:, 1:4].reshape(batch_size, n, 3).contiguous()
            rows = torch.zeros((batch_size, n, rows_ld(feats.shape[-1])), dtype=torch.float32, device=feats.device)
            rows[:, :, :3] = xyz
            rows[:, :, 3:3 + feats.shape[-1]] = feats.reshape(batch_size, n, -1)
        f = self._prepare(rows.device)
        b, n, ld = rows.shape
        lo, hi = self.model_cfg.SAMPLE_RANGE
        cand_rows = rows[:, lo:hi, :].contiguous()
        p = cand_rows.shape[1]

        # vote offsets -> clamp -> vote points (point_head_box6d_vote.py:815-821)
        vote = f['vote']
        spec, kin, wrow0 = [], self.input_channels, 3               # the chain starts at weight row 3 (rows 0..2: coordinates, zero)
        off = torch.empty((b * p, round4(vote[-1][2])), dtype=torch.float32, device=rows.device)
        for li, (w, sh, cout, act) in enumerate(vote):
            spec.append((w, wrow0, sh, kin, cout, act, off if li == len(vote) - 1 else None, 0))
            kin, wrow0 = cout, 0
        if fused.mlp_rows_eligible(self.input_channels, [spec]):    # vote FC stack in one launch (csrc/mlp_rows.hip)
            fused.mlp_rows(cand_rows.view(b * p, ld), 3, [spec])
        else:
            off = run_chain(cand_rows, f['vote'])                   # (B*P, 4), cols 0..2 valid
        vote_xyz = torch.empty((b, p, 3), dtype=torch.float32, device=rows.device)
        off_clamped = torch.empty((b * p, 3), dtype=torch.float32, device=rows.device)
        fused.vote_points(off, cand_rows, self.vote_cfg.MAX_TRANSLATION_RANGE, vote_xyz, off_clamped)

        # SA layer around the votes, then the FC towers
        _, pooled, _ = self.SA_module.forward_rows(xyz, rows, new_xyz=vote_xyz)
        shared = run_chain(pooled.view(b * p, -1), f['shared'])
        ncls, ncode = f['cls'][-1][2], f['reg'][-1][2]
        point_cls_preds = torch.empty((b * p, ncls), dtype=torch.float32, device=rows.device)
        kshared = f['shared'][-1][2]

        def tower(layers, out):
            spec, kin = [], kshared
            for li, (w, sh, cout, act) in enumerate(layers):
                spec.append((w, 0, sh, kin, cout, act, out if li == len(layers) - 1 else None, 0))
                kin = cout
            return spec
        point_reg_preds = torch.empty((b * p, ncode), dtype=torch.float32, device=rows.device)
        towers = [tower(f['cls'], point_cls_preds), tower(f['reg'], point_reg_preds)]
        if shared.shape[1] == kshared and fused.mlp_rows_eligible(kshared, towers):
            fused.mlp_rows(shared, 0, towers)                       # cls and reg towers: two chains, one launch (csrc/mlp_rows.hip)
        else:
            run_chain(shared, f['cls'], out=point_cls_preds)        # the last layer writes the unpadded (B*P, ncls) logits
            reg = run_chain(shared, f['reg'])
            point_reg_preds = reg[:, :ncode].contiguous() if reg.shape[1] != ncode else reg

        vote_flat = vote_xyz.view(b * p, 3)
        boxes = self.box_coder.decode_torch(point_reg_preds, vote_flat)

        cand4 = fused.with_batch_index(cand_rows, 3)
        batch_dict['batch_index'] = cand4[:, 0]
        batch_dict['point_candidate_coords'] = cand4
        batch_dict['point_vote_coords'] = fused.with_batch_index(vote_xyz, 3)
        batch_dict['vote_offsets'] = off_clamped.view(b, p, 3).permute(0, 2, 1)   # (B, 3, P) view, no copy
        batch_dict['point_cls_scores'] = fused.sigmoid_pow(point_cls_preds, 1.0)
        batch_dict['point_box_preds'] = boxes
        batch_dict['batch_cls_preds'] = point_cls_preds
        batch_dict['batch_box_preds'] = boxes
        batch_dict['cls_preds_normalized'] = False
        batch_dict['point_reg_preds'] = point_reg_preds
        self.forward_ret_dict = {'batch_size': batch_size, 'point_cls_preds': point_cls_preds,
                                 'point_reg_preds': point_reg_preds, 'point_box_preds': boxes}
        return batch_dict
