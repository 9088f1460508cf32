"""Input producer, CPU side: the oracle's restatement of mask + sample_points +
collate against fixtures produced by the reference's own DataProcessor (tests/golden/producer.npz),
the keyed bijections it draws from, and the YAML-facing host logic."""
import os

import numpy as np
import pytest

from oracle import ops as oops

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'producer.npz'))
CASES = ('keep_far', 'any_subset', 'pad_once', 'pad_many')


def run_case(name, seed=11):
    frame = GOLD[name + '_frame']
    n = int(GOLD[name + '_num_points'])
    out, n_in = oops.prepare_points([frame], GOLD['point_cloud_range'], n, seed)
    return frame, n, out, int(n_in[0])


@pytest.mark.parametrize('n', [1, 2, 3, 5, 64, 1000, 16384, 65536, 100003])
def test_perm_is_a_bijection(n):
    p = oops.perm(n, seed=1234, scene=7, purpose=2)
    assert np.array_equal(np.sort(p), np.arange(n, dtype=np.uint32))


def test_perm_depends_on_seed_scene_and_purpose():
    base = oops.perm(4096, 1, 0, 1)
    for args in ((2, 0, 1), (1, 1, 1), (1, 0, 2)):
        other = oops.perm(4096, *args)
        assert (other != base).mean() > 0.99
    # roughly uniform: the first quarter of the slots lands evenly over the range
    assert abs(np.mean(base[:1024]) - 2047.5) < 150


@pytest.mark.parametrize('name', CASES)
def test_selection_rule_matches_reference_dataprocessor(name):
    frame, n, out, n_in = run_case(name)
    ref_in, ref_far, ref_chosen = GOLD[name + '_in_ids'], GOLD[name + '_far_ids'], GOLD[name + '_chosen_ids']
    assert out.shape == (n, 5) and np.all(out[:, 0] == 0)
    ids = out[:, 4].astype(np.int64)
    assert np.array_equal(out[:, 1:], frame[ids])          # rows are copied verbatim
    assert n_in == len(ref_in)                               # mask_points_by_range: x / y only
    assert np.all(np.isin(ids, ref_in))
    ours, theirs = np.bincount(ids, minlength=len(frame)), np.bincount(ref_chosen, minlength=len(frame))
    if name == 'keep_far':      # every far point exactly once, the rest distinct near points
        assert np.array_equal(np.sort(ids[np.isin(ids, ref_far)]), np.sort(ref_far))
        assert ours.max() == 1 and theirs.max() == 1
    elif name == 'any_subset':  # N distinct in-range points
        assert ours.max() == 1 and theirs.max() == 1
    elif name == 'pad_once':    # everything once, N - n_in of them twice
        assert np.array_equal(np.sort(ours[ref_in]), np.sort(theirs[ref_in]))
        assert ours[ref_in].min() == 1 and ours.max() == 2
    else:                       # pad_many: everything at least once
        assert ours[ref_in].min() >= 1 and theirs[ref_in].min() >= 1
    assert ours.sum() == theirs.sum() == n


def test_reproducible_and_seed_dependent():
    _, _, a, _ = run_case('keep_far', seed=5)
    _, _, b, _ = run_case('keep_far', seed=5)
    _, _, c, _ = run_case('keep_far', seed=6)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_shuffle_is_not_the_identity_and_near_choice_is_unbiased():
    frame, n, out, _ = run_case('keep_far')
    ids = out[:, 4].astype(np.int64)
    assert np.mean(np.diff(ids) > 0) < 0.6                   # shuffled, not in frame order
    far = set(GOLD['keep_far_far_ids'].tolist())
    near = np.array([i for i in GOLD['keep_far_in_ids'] if i not in far])
    picked = np.zeros(len(near))
    for seed in range(200):
        o, _ = oops.prepare_points([frame], GOLD['point_cloud_range'], n, seed)
        picked += np.isin(near, o[:, 4].astype(np.int64))
    frac = (n - len(far)) / len(near)
    assert abs(picked.mean() / 200 - frac) < 1e-9           # exactly k near points every time
    assert np.abs(picked / 200 - frac).max() < 0.2          # and no point is favoured


def test_batch_and_scene_ids():
    frames = [GOLD[c + '_frame'] for c in CASES]
    rng = GOLD['point_cloud_range']
    out, n_in = oops.prepare_points(frames, rng, 512, 3, scene_ids=[10, 11, 12, 13])
    assert out.shape == (4 * 512, 5)
    assert np.array_equal(out[:, 0], np.repeat(np.arange(4, dtype=np.float32), 512))
    # a frame is sampled the same way wherever it sits in a batch, given its id
    solo, _ = oops.prepare_points([frames[2]], rng, 512, 3, scene_ids=[12])
    assert np.array_equal(solo[:, 1:], out[2 * 512:3 * 512, 1:])
    other, _ = oops.prepare_points([frames[2]], rng, 512, 3, scene_ids=[99])
    assert not np.array_equal(other[:, 1:], solo[:, 1:])


def test_degenerate_frames():
    rng = GOLD['point_cloud_range']
    empty = np.zeros((0, 4), np.float32)
    outside = np.array([[-5, 0, 0, 1], [100, 0, 0, 2]], np.float32)
    single = np.array([[10, 0, -1, 7]], np.float32)
    out, n_in = oops.prepare_points([empty, outside, single], rng, 8, 0)
    assert n_in.tolist() == [0, 0, 1]
    assert np.all(out[:16, 1:] == 0) and np.array_equal(out[:, 0], np.repeat([0, 1, 2], 8).astype(np.float32))
    assert np.all(out[16:, 1:] == single[0])
    # boundary points are inside (>= / <=), z is not tested (common_utils.py:61-64)
    edge = np.array([[0, -40, 50, 1], [70.4, 40, -50, 2], [70.5, 0, 0, 3]], np.float32)
    out, n_in = oops.prepare_points([edge], rng, 2, 0)
    assert n_in[0] == 2 and set(out[:, 4].tolist()) == {1.0, 2.0}


def test_dataprocessor_reads_the_reference_yaml_section():
    from de6d_amd.pcdet.datasets import DataProcessor
    from de6d_amd.runtime import load_config
    cfg = load_config('kitti_models/det6d_car.yaml').DATA_CONFIG
    dp = DataProcessor(cfg.DATA_PROCESSOR, cfg.POINT_CLOUD_RANGE, training=False, num_point_features=4)
    assert dp.mode == 'test' and dp.num_points == 16384 and dp.mask_range and not dp.shuffle
    assert dp.data_processor_queue == ['mask_points_and_boxes_outside_range', 'sample_points', 'shuffle_points']
    assert DataProcessor(cfg.DATA_PROCESSOR, cfg.POINT_CLOUD_RANGE, training=True).shuffle
    with pytest.raises(NotImplementedError):
        DataProcessor([{'NAME': 'transform_points_to_voxels'}], cfg.POINT_CLOUD_RANGE, training=False)
    with pytest.raises(NotImplementedError):
        DataProcessor([{'NAME': 'sample_points', 'NUM_POINTS': {'train': -1, 'test': -1}}], cfg.POINT_CLOUD_RANGE,
                      training=False).forward_batch([np.zeros((4, 4), np.float32)])
