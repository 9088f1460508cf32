"""Input producer on the GPU (det6d_prepare_points through the C ABI) against the CPU oracle:
bit-exact rows for every branch of the reference's sample_points rule, ragged batches, frame ids,
KITTI-sized frames, and the DataProcessor mirror feeding the detector."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'producer.npz'))
RANGE = GOLD['point_cloud_range']
CASES = ('keep_far', 'any_subset', 'pad_once', 'pad_many')


def hip_prepare(frames, num_points, seed, scene_ids=None):
    from de6d_amd.ops import fused
    from de6d_amd.pcdet.datasets import collate_frames
    raw, offsets, _ = collate_frames(frames)
    ids = None if scene_ids is None else torch.tensor(scene_ids, dtype=torch.int32, device='cuda')
    out, n_in = fused.prepare_points(raw, offsets, RANGE, num_points, seed, scene_ids=ids)
    torch.cuda.synchronize()
    return out.cpu().numpy(), n_in.cpu().numpy()


def lidar_frame(seed, n):
    rng = np.random.default_rng(seed)
    r = rng.gamma(2.0, 12.0, n)
    a = rng.uniform(-np.pi, np.pi, n)
    return np.stack([r * np.cos(a), r * np.sin(a), rng.normal(-1.2, 0.6, n), rng.uniform(0, 1, n)], 1).astype(np.float32)


@pytest.mark.parametrize('name', CASES)
def test_every_branch_bit_exact(oracle_ops, name):
    frame, n = GOLD[name + '_frame'], int(GOLD[name + '_num_points'])
    for seed in (0, 17, 2 ** 40 + 5):
        got, got_n = hip_prepare([frame], n, seed)
        ref, ref_n = oracle_ops.prepare_points([frame], RANGE, n, seed)
        assert np.array_equal(got_n, ref_n)
        assert np.array_equal(got, ref)


def test_ragged_batch_with_frame_ids(oracle_ops):
    frames = [GOLD[c + '_frame'] for c in CASES] + [np.zeros((0, 4), np.float32), np.array([[10, 0, -1, 7]], np.float32)]
    ids = [5, 900, 12, 7, 1, 2 ** 31 - 1]
    got, got_n = hip_prepare(frames, 777, 99, ids)
    ref, ref_n = oracle_ops.prepare_points(frames, RANGE, 777, 99, scene_ids=ids)
    assert np.array_equal(got_n, ref_n) and np.array_equal(got, ref)
    assert np.array_equal(got[:, 0], np.repeat(np.arange(6, dtype=np.float32), 777))


def test_kitti_sized_frames(oracle_ops):
    frames = [lidar_frame(s, n) for s, n in ((1, 123397), (2, 115000), (3, 64), (4, 20000))]
    got, got_n = hip_prepare(frames, 16384, 2024)
    ref, ref_n = oracle_ops.prepare_points(frames, RANGE, 16384, 2024)
    assert np.array_equal(got_n, ref_n) and np.array_equal(got, ref)
    # the selection rule on a real-sized frame: every in-range point beyond 40 m survives
    f = frames[0]
    in_range = (f[:, 0] >= RANGE[0]) & (f[:, 0] <= RANGE[3]) & (f[:, 1] >= RANGE[1]) & (f[:, 1] <= RANGE[4])
    far = in_range & ~(np.linalg.norm(f[:, 0:3], axis=1) < 40.0)
    assert in_range.sum() == got_n[0] > 16384 > far.sum()
    rows = {r.tobytes() for r in got[:16384, 1:]}
    assert all(r.tobytes() in rows for r in f[far])
    assert len(rows) == 16384


def test_dataprocessor_feeds_the_detector():
    from de6d_amd.pcdet.datasets import DataProcessor
    from de6d_amd.runtime import load_config, build_model
    cfg = load_config('synthetic_models/det6d_tiny.yaml')
    dc = cfg.DATA_CONFIG
    dp = DataProcessor(dc.DATA_PROCESSOR, dc.POINT_CLOUD_RANGE, training=False, seed=3)
    model = build_model(cfg, seed=21, device='cuda')
    frames = [lidar_frame(10 + i, 30000 + 1000 * i) for i in range(3)]
    batch = dp.forward_batch(frames, frame_ids=[100, 101, 102])
    assert batch['points'].shape == (3 * dp.num_points, 5) and batch['batch_size'] == 3
    with torch.no_grad():
        pred, _ = model(dict(batch))
    assert len(pred) == 3
    # sharding-invariant: frame 101 alone is sampled, and therefore detected, identically
    solo = dp.forward_batch([frames[1]], frame_ids=[101])
    n = dp.num_points
    assert torch.equal(solo['points'][:, 1:], batch['points'][n:2 * n, 1:])
    with torch.no_grad():
        pred_solo, _ = model(dict(solo))
    assert torch.equal(pred_solo[0]['pred_boxes'], pred[1]['pred_boxes'])
    # single-frame reference interface
    one = dp.forward({'points': frames[1], 'frame_index': 101})
    assert torch.equal(one['points'], solo['points'][:, 1:])
