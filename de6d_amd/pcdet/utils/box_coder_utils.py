"""Box coders of core/pcdet/utils/box_coder_utils.py used on the Det6D path.

`PointBinResidual6DCoder` (:546-737): code = 6 offsets + `angle_bin_num` yaw-bin logits +
`angle_bin_num` yaw residuals + (pitch logit, pitch residual) when ground_aware.  decode_torch
runs as one HIP kernel (csrc/points.hip: decode_boxes_kernel)."""
import numpy as np
import torch

from ..ops_backend import fused


class PointBinResidual6DCoder(object):
    def __init__(self, use_mean_size=True, ground_aware=True, angle_bin_num=12, minus=False, threshold=10,
                 factor=45, **kwargs):
        self.ground_aware = ground_aware
        self.angle_bin_num = angle_bin_num
        self.use_mean_size = use_mean_size
        self.minus = minus
        if self.use_mean_size:
            # the reference forward passes pred_classes=None into the decode (point_head_box6d_vote.py:864),
            # which would dereference None at box_coder_utils.py:655: the mode is unusable there too
            raise NotImplementedError("use_mean_size=True is not supported by the Det6D head (SURVEY.md a11)")
        self.code_size = 6 + 2 * angle_bin_num + (2 if ground_aware else 1)
        self.ground_threshold = np.deg2rad(threshold)
        self.ground_factor = np.deg2rad(factor)

    def decode_torch(self, box_encodings, points, pred_classes=None):
        """(N, code_size [+ extras]), (N, 3) device tensors -> (N, 9 [+ extras]) boxes
        [x, y, z, dx, dy, dz, rz, ry, rx]"""
        code = box_encodings.contiguous()
        pts = points.contiguous()
        boxes = fused.decode_boxes(code, pts, self.angle_bin_num, self.ground_aware, self.minus,
                                   np.float32(self.ground_threshold), np.float32(self.ground_factor))
        if code.shape[-1] > self.code_size:
            boxes = torch.cat([boxes, code[:, self.code_size:]], dim=-1)
        return boxes

    # ---- training-side encoders (pure torch; not on the inference hot path) -------------
    def encode_rz_torch(self, angle):
        two_pi = np.pi * 2.0
        per_bin = two_pi / float(self.angle_bin_num)
        shifted = torch.remainder(torch.remainder(angle, two_pi) + per_bin / 2.0, two_pi)
        bin_f = (shifted / per_bin).floor()
        onehot = bin_f.new_zeros(*bin_f.shape, self.angle_bin_num)
        onehot.scatter_(-1, bin_f.unsqueeze(-1).long(), 1.0)
        res = (shifted - (bin_f * per_bin + per_bin / 2.0)) / per_bin
        return onehot, onehot * res.unsqueeze(-1)

    def encoder_rxry_torch(self, rx, ry):
        pitch = ry
        if not self.ground_aware:
            return pitch,
        res = torch.zeros_like(pitch)
        if self.minus:
            flag = torch.abs(pitch) > self.ground_threshold
            res[flag] = pitch[flag] / self.ground_factor
        else:
            flag = pitch < -self.ground_threshold
            res[flag] = (-self.ground_threshold - pitch[flag]) / self.ground_factor
        return flag.long(), res

    def encode_torch(self, gt_boxes, points, gt_classes=None):
        assert gt_boxes.shape[-1] >= 9, 'gt_boxes shape: %s' % str(gt_boxes.shape)
        gt_boxes[:, 3:6] = torch.clamp_min(gt_boxes[:, 3:6], min=1e-5)
        xg, yg, zg, dxg, dyg, dzg, rzg, ryg, rxg, *cgs = torch.split(gt_boxes, 1, dim=-1)
        xa, ya, za = torch.split(points, 1, dim=-1)
        rz_cls, rz_reg = self.encode_rz_torch(rzg.squeeze(-1))
        return torch.cat([xg - xa, yg - ya, zg - za, torch.log(dxg), torch.log(dyg), torch.log(dzg),
                          rz_cls, rz_reg, *self.encoder_rxry_torch(rxg, ryg)], dim=-1)
