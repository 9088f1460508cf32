"""Python op API of core/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py (same public
names, argument orders and tensor layouts), bound to libdet6d_hip through
``de6d_amd.ops.pointnet2_batch_hip`` instead of ``pointnet2_batch_cuda``.

Outputs are allocated here exactly as the reference does (int32 indices, `temp` filled with
1e10, ball-query `idx` zero-filled: pointnet2_utils.py:25-26,294,323-324,347-348).
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from ....ops_backend import pointnet2_batch_hip as pointnet2


def _i32(*shape, device):
    return torch.empty(shape, dtype=torch.int32, device=device)


def _f32(*shape, device):
    return torch.empty(shape, dtype=torch.float32, device=device)


class FarthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        assert xyz.is_contiguous()
        b, n, _ = xyz.size()
        out = _i32(b, npoint, device=xyz.device)
        temp = torch.full((b, n), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.farthest_point_sampling_wrapper(b, n, npoint, xyz, temp, out)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


class FurthestPointSamplingWeights(Function):
    @staticmethod
    def forward(ctx, xyz, weights, npoint):
        assert xyz.is_contiguous() and weights.is_contiguous()
        b, n, _ = xyz.size()
        out = _i32(b, npoint, device=xyz.device)
        temp = torch.full((b, n), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.furthest_point_sampling_weights_wrapper(b, n, npoint, xyz, weights, temp, out)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad=None):
        return None, None, None


furthest_point_sample_weights = FurthestPointSamplingWeights.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        assert features.is_contiguous() and idx.is_contiguous()
        b, npoint = idx.size()
        _, c, n = features.size()
        out = _f32(b, c, npoint, device=features.device)
        pointnet2.gather_points_wrapper(b, c, n, npoint, features, idx, out)
        ctx.for_backwards = (idx, c, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, c, n = ctx.for_backwards
        b, npoint = idx.size()
        grad_features = torch.zeros((b, c, n), dtype=torch.float32, device=grad_out.device)
        pointnet2.gather_points_grad_wrapper(b, c, n, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        assert unknown.is_contiguous() and known.is_contiguous()
        b, n, _ = unknown.size()
        m = known.size(1)
        dist2 = _f32(b, n, 3, device=unknown.device)
        idx = _i32(b, n, 3, device=unknown.device)
        pointnet2.three_nn_wrapper(b, n, m, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        assert features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
        b, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        out = _f32(b, c, n, device=features.device)
        pointnet2.three_interpolate_wrapper(b, c, m, n, features, idx, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        b, c, n = grad_out.size()
        grad_features = torch.zeros((b, c, m), dtype=torch.float32, device=grad_out.device)
        pointnet2.three_interpolate_grad_wrapper(b, c, n, m, grad_out.contiguous(), idx, weight, grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        assert features.is_contiguous() and idx.is_contiguous()
        b, nfeatures, nsample = idx.size()
        _, c, n = features.size()
        out = _f32(b, c, nfeatures, nsample, device=features.device)
        pointnet2.group_points_wrapper(b, c, n, nfeatures, nsample, features, idx, out)
        ctx.for_backwards = (idx, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, n = ctx.for_backwards
        b, c, npoint, nsample = grad_out.size()
        grad_features = torch.zeros((b, c, n), dtype=torch.float32, device=grad_out.device)
        pointnet2.group_points_grad_wrapper(b, c, n, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        assert new_xyz.is_contiguous() and xyz.is_contiguous()
        b, n, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.zeros((b, npoint, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(b, n, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class BallQueryCnt(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        assert new_xyz.is_contiguous() and xyz.is_contiguous()
        b, n, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx_cnt = torch.zeros((b, npoint), dtype=torch.int32, device=xyz.device)
        idx = torch.zeros((b, npoint, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_cnt_wrapper(b, n, npoint, radius, nsample, new_xyz, xyz, idx_cnt, idx)
        ctx.mark_non_differentiable(idx_cnt, idx)
        return idx_cnt, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None


ball_query_cnt = BallQueryCnt.apply


class BallQueryDilated(Function):
    @staticmethod
    def forward(ctx, radius_in, radius_out, nsample, xyz, new_xyz):
        assert new_xyz.is_contiguous() and xyz.is_contiguous()
        b, n, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx_cnt = torch.zeros((b, npoint), dtype=torch.int32, device=xyz.device)
        idx = torch.zeros((b, npoint, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_dilated_wrapper(b, n, npoint, radius_in, radius_out, nsample, new_xyz, xyz,
                                             idx_cnt, idx)
        ctx.mark_non_differentiable(idx_cnt, idx)
        return idx_cnt, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None


ball_query_dilated = BallQueryDilated.apply


def _group(xyz, new_xyz, features, idx, use_xyz):
    """grouped [xyz - centre, features] tensor (B, 3 + C, npoint, nsample), pointnet2_utils.py:445-461"""
    rel = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
    rel = rel - new_xyz.transpose(1, 2).unsqueeze(-1)
    if features is None:
        assert use_xyz, "Cannot have not features and not use xyz as a feature!"
        return rel
    feats = grouping_operation(features, idx)
    return torch.cat([rel, feats], dim=1) if use_xyz else feats


class QueryAndGroup(nn.Module):
    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        return _group(xyz, new_xyz, features, idx, self.use_xyz)


class QueryWithCntAndGroup(nn.Module):
    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx_cnt, idx = ball_query_cnt(self.radius, self.nsample, xyz, new_xyz)
        return idx_cnt, _group(xyz, new_xyz, features, idx, self.use_xyz)


class QueryAndGroupDilated(nn.Module):
    def __init__(self, radius_in, radius_out, nsample, use_xyz=True):
        super().__init__()
        self.radius_in, self.radius_out, self.nsample, self.use_xyz = radius_in, radius_out, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx_cnt, idx = ball_query_dilated(self.radius_in, self.radius_out, self.nsample, xyz, new_xyz)
        return idx_cnt, _group(xyz, new_xyz, features, idx, self.use_xyz)


class GroupAll(nn.Module):
    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
