"""SlopeAug on the GPU: det6d_make_slope / det6d_boxes9_corners == the C oracle bit for bit, and the
augmentor mirror (seeded np.random, same draws as the reference) reproduces the reference's own
output (tests/golden/slope.npz) within the stated tolerance, plain and smooth."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'slope.npz'))
PARAMS = tuple(GOLD['params'])


def test_kernels_equal_c_oracle(oracle_ops):
    from de6d_amd.ops import fused
    from de6d_amd.pcdet.datasets.augmentor.augmentor_utils import slope_params
    for c in range(2):
        prm = slope_params(GOLD['out_%d_pivot' % c], GOLD['out_%d_angle' % c])
        pts = np.tile(GOLD['in_%d_points' % c], (8, 1))                     # several workgroups
        b9 = np.concatenate([GOLD['in_%d_boxes' % c].astype(np.float64), np.zeros((24, 2))], 1)
        b9[:, 7] = np.linspace(-4, 4, 24)                                      # exercises the angle wrap
        d_pts, d_box = torch.from_numpy(pts).cuda(), torch.from_numpy(b9).cuda()
        fused.make_slope(d_pts, d_box, prm)
        ref_pts, ref_box = oracle_ops.make_slope(pts, b9, prm)
        assert np.array_equal(d_pts.cpu().numpy(), ref_pts)
        assert np.array_equal(d_box.cpu().numpy(), ref_box)
        corners = fused.boxes9_corners(d_box).cpu().numpy()
        # device cos/sin (ocml) vs libm may differ in the last bit: 1e-12 on metre-scale corners
        assert np.allclose(corners, oracle_ops.boxes9_corners(ref_box), atol=1e-12, rtol=0)


@pytest.mark.parametrize('c', range(4))
def test_mirror_reproduces_reference(c):
    from de6d_amd.pcdet.datasets.augmentor import augmentor_utils
    from de6d_amd.ops import fused
    np.random.seed(int(GOLD['in_%d_seed' % c]))
    gt, pts, pivot, angle = augmentor_utils.random_global_make_slope(
        GOLD['in_%d_boxes' % c].copy(), GOLD['in_%d_points' % c].copy(), params=PARAMS, smooth=bool(GOLD['in_%d_smooth' % c]))
    assert np.array_equal(pivot, GOLD['out_%d_pivot' % c]) and np.array_equal(angle, GOLD['out_%d_angle' % c])   # same draws
    want = GOLD['out_%d_points' % c]
    assert np.allclose(pts, want, atol=2e-5, rtol=0) and (pts != want).mean() < 0.02     # float32, <= 1-2 ulp at 80 m
    assert np.allclose(gt, GOLD['out_%d_boxes' % c], atol=1e-6, rtol=0)
    corners = fused.boxes9_corners(torch.from_numpy(gt).cuda()).cpu().numpy()
    assert np.allclose(corners, GOLD['out_%d_corners' % c], atol=1e-6, rtol=0)


def test_data_augmentor_entry_and_tensor_inputs():
    from de6d_amd.pcdet.datasets.augmentor.data_augmentor import DataAugmentor
    cfg = [{'NAME': 'random_make_slope_in_scene', 'PROB': 1.0, 'SMOOTH': False,
            'SLOPE_DISTANCE': {'MEAN': 20, 'VAR': 10}, 'SLOPE_ANGLE': {'MEAN': 20, 'VAR': 8}}]
    aug = DataAugmentor(None, cfg, ['Car'])
    pts, boxes = GOLD['in_0_points'].copy(), GOLD['in_0_boxes'].copy()
    np.random.seed(5)
    out = aug.forward({'points': pts.copy(), 'gt_boxes': boxes.copy()})
    assert out['gt_boxes'].shape == (24, 9) and out['points'].shape == pts.shape and (out['points'] != pts).any()
    np.random.seed(5)
    out_t = aug.forward({'points': torch.from_numpy(pts).cuda(), 'gt_boxes': boxes.copy()})   # device-resident points
    assert torch.is_tensor(out_t['points']) and np.array_equal(out_t['points'].cpu().numpy(), out['points'])
    skip = DataAugmentor(None, [dict(cfg[0], PROB=0.0)], ['Car']).forward({'points': pts.copy(), 'gt_boxes': boxes.copy()})
    assert np.array_equal(skip['points'], pts) and skip['gt_boxes'].shape == (24, 9) and np.all(skip['gt_boxes'][:, 7:] == 0)
    with pytest.raises(NotImplementedError):
        DataAugmentor(None, [{'NAME': 'gt_sampling'}], ['Car'])
