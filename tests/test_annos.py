"""Output consumer, CPU side: the NumPy restatement of the reference's
generate_prediction_dicts against fixtures written by the reference itself (tests/golden/annos.npz),
and the scalar C oracle of det6d_kitti_annos against that restatement."""
import os

import numpy as np
import pytest

from oracle import annos as oannos
from oracle import ops as oops

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'annos.npz'))
NAMES = ['Car', 'Pedestrian', 'Cyclist']
N_FRAMES = int(GOLD['n_frames'])
TOL = 1e-4   # float32 BLAS-order / libm differences between NumPy and the fma-chain kernels (include/det6d_ops.h)


def frame_inputs(i):
    calib = oannos.Calib(GOLD['in_%d_P2' % i], GOLD['in_%d_R0' % i], GOLD['in_%d_Tr_velo2cam' % i])
    return GOLD['in_%d_boxes' % i], GOLD['in_%d_scores' % i], GOLD['in_%d_labels' % i], calib, GOLD['in_%d_image_shape' % i]


def packed_calib(calib, image_shape):
    out = np.zeros(28, np.float32)
    out[0:12] = np.dot(calib.V2C.T, calib.R0.T).reshape(-1)
    out[12:24] = calib.P2.reshape(-1)
    out[24], out[25] = image_shape[0], image_shape[1]
    return out


@pytest.mark.parametrize('tag,ncol', [('kitti', 7), ('sloped', 9)])
def test_numpy_restatement_equals_reference(tag, ncol):
    for i in range(N_FRAMES):
        boxes, scores, labels, calib, shape = frame_inputs(i)
        d = oannos.prediction_dict(boxes[:, :ncol].copy(), scores, labels, calib, shape, NAMES, sloped=tag == 'sloped')
        for key in ('alpha', 'bbox', 'dimensions', 'location', 'rotation_y', 'score', 'boxes_lidar') + \
                (('pitch', 'roll') if tag == 'sloped' else ()):
            want = GOLD['%s_%d_%s' % (tag, i, key)]
            assert np.asarray(d[key]).shape == want.shape, key
            assert np.array_equal(np.asarray(d[key]), want), (key, i)
        assert [str(n) for n in d['name']] == [str(n) for n in GOLD['%s_%d_name' % (tag, i)]]
        txt = str(GOLD['%s_%d_txt' % (tag, i)])
        assert '\n'.join(oannos.kitti_lines(d, sloped=tag == 'sloped')) == txt.rstrip('\n')


def test_c_oracle_within_tolerance_of_numpy():
    for i in range(N_FRAMES):
        boxes, scores, labels, calib, shape = frame_inputs(i)
        if len(boxes) == 0:
            continue
        got = oops.kitti_annos(boxes, np.zeros(len(boxes), np.int32), packed_calib(calib, shape)[None])
        d = oannos.prediction_dict(boxes.copy(), scores, labels, calib, shape, NAMES, sloped=True)
        assert np.allclose(got[:, 0:3], d['location'], atol=TOL, rtol=0)
        assert np.allclose(got[:, 3:6], d['dimensions'], atol=0, rtol=0)
        assert np.allclose(got[:, 6], d['rotation_y'], atol=1e-6, rtol=0)
        assert np.allclose(got[:, 7:11], d['bbox'], atol=5e-3, rtol=1e-5)      # pixels, values up to 1242
        assert np.allclose(got[:, 11], d['alpha'], atol=TOL, rtol=0)


def test_image_box_clipping_and_no_clipping():
    boxes, _, _, calib, shape = frame_inputs(0)
    free = packed_calib(calib, (0, 0))
    clip = packed_calib(calib, shape)
    a = oops.kitti_annos(boxes, np.zeros(len(boxes), np.int32), free[None])
    b = oops.kitti_annos(boxes, np.zeros(len(boxes), np.int32), clip[None])
    assert (a[:, 7] < 0).any() or (a[:, 9] > shape[1] - 1).any()       # something sticks out of the image
    assert b[:, 7:11].min() >= 0 and b[:, 9].max() <= shape[1] - 1 and b[:, 10].max() <= shape[0] - 1
    assert np.array_equal(np.clip(a[:, 7], 0, shape[1] - 1), b[:, 7])


def test_calibration_mirror_parses_kitti_file(tmp_path):
    from de6d_amd.pcdet.utils.calibration_kitti import Calibration, get_calib_from_file
    P2, R0, V2C = GOLD['in_0_P2'], GOLD['in_0_R0'], GOLD['in_0_Tr_velo2cam']
    rows = [('P0', P2), ('P1', P2), ('P2', P2), ('P3', P2), ('R0_rect', R0), ('Tr_velo_to_cam', V2C), ('Tr_imu_to_velo', V2C)]
    path = tmp_path / '000007.txt'
    path.write_text('\n'.join('%s: %s' % (k, ' '.join('%.12e' % v for v in m.reshape(-1))) for k, m in rows) + '\n')
    parsed = get_calib_from_file(str(path))
    assert np.array_equal(parsed['P2'], P2) and np.array_equal(parsed['R0'], R0) and np.array_equal(parsed['Tr_velo2cam'], V2C)
    c = Calibration(str(path))
    ref = oannos.Calib(P2, R0, V2C)
    pts = np.random.default_rng(0).uniform(-20, 40, (50, 3)).astype(np.float32)
    assert np.array_equal(c.lidar_to_rect(pts), ref.lidar_to_rect(pts))
    assert np.array_equal(c.rect_to_img(c.lidar_to_rect(pts))[0], ref.rect_to_img(ref.lidar_to_rect(pts)))
    assert np.array_equal(c.packed((375, 1242)), packed_calib(ref, (375, 1242)))
