"""HIP path of the feature-propagation branch (FP_MLPS -> PointnetFPModule: three_nn + three_interpolate + shared MLP) and of
boxes_iou3d_gpu, against the reference-written fixture tests/golden/fp.npz and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from tests.test_fp import G, build_fp_backbone, build_fp_module, fp_backbone_cfg, fp_inputs

pytestmark = pytest.mark.gpu


def test_backbone_with_feature_propagation_hip(oracle_ops):
    from oracle import model as omodel
    z = np.load(os.path.join(G, 'fp.npz'))
    net = build_fp_backbone(z, device='cuda')
    pts, b = fp_inputs(z)
    bd = {'batch_size': b, 'points': torch.from_numpy(pts).cuda()}
    with torch.no_grad():
        bd = net(bd)
    for lvl in range(3):
        np.testing.assert_array_equal(bd['point_coords_list'][lvl].cpu().numpy(), z['point_coords_list_%d' % lvl])
    np.testing.assert_array_equal(bd['point_coords'].cpu().numpy(), z['point_coords'])
    feats = bd['point_features'].cpu().numpy()
    np.testing.assert_allclose(feats, z['point_features'], atol=1e-4)          # the reference's own Python
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    ref = omodel.backbone_forward(fp_backbone_cfg(), sd, pts, b, prefix='')
    np.testing.assert_allclose(feats, ref['point_features'], atol=2e-6)        # the oracle (weights: torch vs numpy division)


def test_fp_module_alone_hip():
    z = np.load(os.path.join(G, 'fp.npz'))
    fp = build_fp_module(z, device='cuda')
    with torch.no_grad():
        y = fp(torch.from_numpy(z['fp_unknown']).cuda(), torch.from_numpy(z['fp_known']).cuda(), None,
               torch.from_numpy(z['fp_known_feats']).cuda())
    np.testing.assert_allclose(y.cpu().numpy(), z['fp_out'], atol=1e-4)
    # known = None: the single global feature is broadcast (pointnet2_modules.py:156-157)
    with torch.no_grad():
        y1 = fp(torch.from_numpy(z['fp_unknown']).cuda(), None, None, torch.from_numpy(z['fp_known_feats'][:, :, :1]).cuda())
    assert y1.shape == (2, 8, 300) and torch.equal(y1[:, :, 0], y1[:, :, 299])


def test_boxes_iou3d_gpu_against_reference():
    from de6d_amd.pcdet.ops.iou3d_nms import iou3d_nms_utils
    z = np.load(os.path.join(G, 'fp.npz'))
    a, b = torch.from_numpy(z['iou3d_a']).cuda(), torch.from_numpy(z['iou3d_b']).cuda()
    np.testing.assert_allclose(iou3d_nms_utils.boxes_iou3d_gpu(a, b).cpu().numpy(), z['iou3d'], atol=2e-5)
    np.testing.assert_allclose(iou3d_nms_utils.boxes_iou_bev(a, b).cpu().numpy(), z['iou_bev'], atol=2e-5)
