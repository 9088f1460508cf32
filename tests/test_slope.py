"""SlopeAug, CPU side: the NumPy/scipy restatement against fixtures from the
reference's own random_global_make_slope / boxes3d_to_corners_3d (tests/golden/slope.npz), the scalar
C oracle of det6d_make_slope / det6d_boxes9_corners against that restatement, and the host-side
parameter helpers of the mirror."""
import os

import numpy as np
import pytest

from oracle import ops as oops
from oracle import slope as oslope

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'slope.npz'))
PARAMS = tuple(GOLD['params'])
TOL = 1e-6   # fma-chain dots vs BLAS, float32 re-rounding of the points (include/det6d_ops.h)


def case_inputs(c):
    return GOLD['in_%d_boxes' % c].copy(), GOLD['in_%d_points' % c].copy(), int(GOLD['in_%d_seed' % c]), bool(GOLD['in_%d_smooth' % c])


@pytest.mark.parametrize('c', range(4))
def test_numpy_restatement_equals_reference(c):
    boxes, points, seed, smooth = case_inputs(c)
    np.random.seed(seed)
    gt, pts, pivot, angle = oslope.random_global_make_slope(boxes, points, params=PARAMS, smooth=smooth)
    assert np.array_equal(pivot, GOLD['out_%d_pivot' % c]) and np.array_equal(angle, GOLD['out_%d_angle' % c])
    assert np.array_equal(pts, GOLD['out_%d_points' % c])
    assert np.array_equal(gt, GOLD['out_%d_boxes' % c])
    assert np.array_equal(oslope.boxes9_corners(gt.copy()), GOLD['out_%d_corners' % c])


def host_params(pivot, angle):
    from scipy.spatial.transform import Rotation
    k = angle[1] / (angle[0] + 1e-6)
    e = Rotation.from_rotvec(angle).as_euler('XYZ')
    return np.concatenate([pivot, Rotation.from_rotvec(angle).as_matrix().reshape(-1),
                           [k, np.sign(k * (0 - pivot[0]) + pivot[1]), e[1], e[0]]])


@pytest.mark.parametrize('c', [0, 1])
def test_c_oracle_within_tolerance_of_reference(c):
    boxes, points, _, _ = case_inputs(c)
    b9 = np.concatenate([boxes.astype(np.float64), np.zeros((len(boxes), 2))], 1)
    pts, bx = oops.make_slope(points, b9, host_params(GOLD['out_%d_pivot' % c], GOLD['out_%d_angle' % c]))
    assert np.allclose(pts, GOLD['out_%d_points' % c], atol=1e-5, rtol=0)   # float32 coordinates up to 80 m: 1 ulp = 8e-6
    assert (pts != GOLD['out_%d_points' % c]).mean() < 0.02                  # and almost all of them bit-identical
    assert np.allclose(bx, GOLD['out_%d_boxes' % c], atol=TOL, rtol=0)
    assert np.allclose(oops.boxes9_corners(bx), GOLD['out_%d_corners' % c], atol=TOL, rtol=0)


def test_slope_rule_properties():
    boxes, points, _, _ = case_inputs(0)
    pivot, angle = GOLD['out_0_pivot'], GOLD['out_0_angle']
    out = GOLD['out_0_points']
    moved = (out != points).any(1)
    k = angle[1] / (angle[0] + 1e-6)
    beyond = np.sign(k * (points[:, 0] - pivot[0]) + pivot[1] - points[:, 1]) != np.sign(k * (0 - pivot[0]) + pivot[1])
    assert not moved[~beyond].any() and moved[beyond].mean() > 0.99
    # rigid: distances to the pivot are preserved, intensity untouched
    d0 = np.linalg.norm(points[beyond, :3] - pivot, axis=1)
    d1 = np.linalg.norm(out[beyond, :3] - pivot, axis=1)
    assert np.allclose(d0, d1, atol=1e-4) and np.array_equal(points[:, 3], out[:, 3])
    gt = GOLD['out_0_boxes']
    assert gt.shape[1] == 9 and np.all(np.abs(gt[:, 6:9]) <= np.pi)
    assert ((gt[:, 7] != 0) | (gt[:, 8] != 0)).any()


def test_mirror_parameter_helpers_and_rng_order():
    import importlib
    aug = importlib.import_module('de6d_amd.pcdet.datasets.augmentor.augmentor_utils')
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0)
    for _ in range(100):
        v = rng.normal(0, 0.4, 3)
        assert np.allclose(aug.rotvec_to_matrix(v), Rotation.from_rotvec(v).as_matrix(), atol=1e-14)
        assert np.allclose(aug.matrix_to_euler_XYZ(aug.rotvec_to_matrix(v)), Rotation.from_rotvec(v).as_euler('XYZ'), atol=1e-13)
    assert np.allclose(aug.slope_params(GOLD['out_0_pivot'], GOLD['out_0_angle']),
                       host_params(GOLD['out_0_pivot'], GOLD['out_0_angle']), atol=1e-13)
    assert np.array_equal(aug.rotvec_to_matrix(np.zeros(3)), np.eye(3))
