"""KITTI / SlopedKITTI evaluator, CPU side: the evaluator mirror driven by the CPU-oracle
backend against what the reference's own eval.py produced on the same annotations (executed as plain
Python, tests/golden/make_golden.py: gen_eval) — overlaps of every metric, the 41-point precision tables,
the AP dictionary and the printed report — plus unit checks of the host-side pieces."""
import numpy as np
import pytest

from oracle import ops as oops
from tests.eval_util import CLASSES, GOLD, N_FRAMES, annos


def ev():
    from de6d_amd.pcdet.datasets.kitti.kitti_object_eval_python import eval as module
    return module


@pytest.mark.parametrize('metric,tol', [(0, 1e-7), (1, 2e-5), (2, 2e-5), (3, 1e-7)])
def test_overlaps_match_reference(metric, tol):
    """tolerance: the reference fixture comes from plain-Python execution of the numba code (NumPy scalar
    typing instead of numba's), and the rotated IoU is float32 geometry"""
    lay = ev().SplitLayout(annos('gt'), annos('dt'), metrics=(0, 1, 2, 3))
    ov = oops.EvalBackend(lay).overlaps_host(metric)
    for f in range(N_FRAMES):
        want = GOLD['sloped_overlap_m%d_f%d' % (metric, f)]
        got = ov[lay.pair_off[f]:lay.pair_off[f + 1]].reshape(want.shape)
        assert np.allclose(got, want, atol=tol, rtol=0), (metric, f)
        if metric < 3:
            assert np.allclose(got, GOLD['kitti_overlap_m%d_f%d' % (metric, f)], atol=tol, rtol=0)


def test_kitti_report_tables_and_dictionary_equal_reference():
    gts, dts = annos('gt'), annos('dt')
    detail = {}
    text, ret = ev().get_official_eval_result(gts, dts, CLASSES, PR_detail_dict=detail,
                                              backend=oops.EvalBackend(ev().SplitLayout(gts, dts)))
    assert text == str(GOLD['kitti_report'])
    for key in ('bbox', 'bev', '3d', 'aos'):
        want = GOLD['kitti_precision_' + key]
        assert detail[key].shape == want.shape and np.allclose(detail[key], want, atol=1e-12, rtol=0, equal_nan=True)
    assert sorted(ret) == list(GOLD['kitti_ret_keys'])
    assert np.allclose([ret[k] for k in sorted(ret)], GOLD['kitti_ret_vals'], atol=1e-9, rtol=0)
    assert max(ret.values()) > 20       # the synthetic split is not degenerate


def test_sloped_report_tables_and_dictionary_equal_reference():
    gts, dts = annos('gt'), annos('dt')
    detail = {}
    text, ret = ev().get_slopedkitti_eval_result(gts, dts, CLASSES, PR_detail_dict=detail,
                                                 backend=oops.EvalBackend(ev().SplitLayout(gts, dts, metrics=(0, 1, 2, 3))))
    assert text == str(GOLD['sloped_report'])          # includes CAP / ATS / ASS / AOS / ODS of the "all" level
    for key in ('bbox', 'bev', '3d', 'aos', '3dctr'):
        want = GOLD['sloped_precision_' + key]
        assert detail[key].shape == want.shape and np.allclose(detail[key], want, atol=1e-12, rtol=0, equal_nan=True)
    assert sorted(ret) == list(GOLD['sloped_ret_keys'])
    assert np.allclose([ret[k] for k in sorted(ret)], GOLD['sloped_ret_vals'], atol=1e-9, rtol=0)


def test_get_thresholds_and_map_helpers():
    m = ev()
    scores = np.array([0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2])
    thr = m.get_thresholds(scores.copy(), num_gt=8, num_sample_pts=5)   # recall positions 0, .25, .5, .75, 1
    assert thr == [0.9, 0.8, 0.6, 0.4, 0.2]
    assert m.get_thresholds(np.zeros(0), 5) == []
    prec = np.zeros((2, 41)); prec[0] = 1.0; prec[1, :21] = 0.5
    assert np.allclose(m.get_mAP(prec), [100.0, 50.0 * 6 / 11]) and np.allclose(m.get_mAP_R40(prec), [100.0, 25.0])


def test_clean_data_rules():
    m = ev()
    gt = dict(name=np.array(['Car', 'Van', 'Car', 'Pedestrian', 'DontCare', 'Car']),
              bbox=np.array([[0, 0, 50, 60], [0, 0, 50, 60], [0, 0, 50, 30], [0, 0, 20, 60], [5, 5, 9, 9], [0, 0, 50, 60.0]]),
              occluded=np.array([0, 0, 0, 0, 0, 2]), truncated=np.array([0, 0, 0, 0, 0, 0.0]))
    dt = dict(name=np.array(['Car', 'Car', 'Cyclist']), bbox=np.array([[0, 0, 10, 50], [0, 0, 10, 30], [0, 0, 10, 50]], np.float32))
    n_valid, ign_gt, ign_dt, dc = m.clean_data(gt, dt, 0, 0)
    assert n_valid == 1 and ign_gt.tolist() == [0, 1, 1, -1, -1, 1] and ign_dt.tolist() == [0, 1, -1]
    assert dc.tolist() == [[5, 5, 9, 9]]
    n_valid, ign_gt, _, _ = m.clean_data(gt, dt, 0, 2)            # hard: 25 px, occlusion 2 allowed
    assert n_valid == 3 and ign_gt.tolist() == [0, 1, 0, -1, -1, 0]
    n_valid, ign_gt, ign_dt, _ = m.clean_data(gt, dt, 1, 3)       # pedestrians, SlopedKITTI "all" level
    assert n_valid == 1 and ign_gt.tolist() == [-1, -1, -1, 0, -1, -1] and ign_dt.tolist() == [-1, -1, -1]


def test_label_reader_round_trip(tmp_path):
    from de6d_amd.pcdet.datasets.kitti.kitti_object_eval_python import kitti_common
    (tmp_path / '000003.txt').write_text('Car 0.00 1 1.5500 10.00 20.00 110.00 90.00 1.5000 1.6000 3.9000 1.0000 1.6000 20.0000 -1.5600\n'
                                         'DontCare -1 -1 -10 5.0 6.0 7.0 8.0 -1 -1 -1 -1000 -1000 -1000 -10\n')
    (tmp_path / '000004.txt').write_text('Car -1 -1 0.1000 1.0 2.0 3.0 4.0 1.5 1.6 3.9 1.0 1.6 20.0 0.2000 -0.3000 0.0000 0.8765\n')
    (tmp_path / '000005.txt').write_text('')
    a3, a4, a5 = kitti_common.get_label_annos(tmp_path)
    assert a3['name'].tolist() == ['Car', 'DontCare'] and a3['occluded'].dtype == np.int64
    assert a3['dimensions'][0].tolist() == [3.9, 1.5, 1.6] and a3['score'].tolist() == [0.0, 0.0]
    assert a4['pitch'].tolist() == [-0.3] and a4['roll'].tolist() == [0.0] and a4['score'].tolist() == [0.8765]
    assert len(a5['name']) == 0 and a5['bbox'].shape == (0, 4)
    assert [len(a['name']) for a in kitti_common.get_label_annos(tmp_path, [4])] == [1]
