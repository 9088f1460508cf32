"""Output consumer on the GPU: det6d_kitti_annos == the C oracle bit for bit, the dataset mirrors
reproduce the reference's annotation dicts / label files (fixtures from the reference itself) within
the stated float32 tolerance, and eval_one_epoch runs the whole loop."""
import logging
import os
import pickle
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'annos.npz'))
NAMES = ['Car', 'Pedestrian', 'Cyclist']
N_FRAMES = int(GOLD['n_frames'])
TOL = 1e-4


def batch_and_preds(ncol):
    from de6d_amd.pcdet.utils.calibration_kitti import Calibration
    calibs = [Calibration({'P2': GOLD['in_%d_P2' % i], 'R0': GOLD['in_%d_R0' % i], 'Tr_velo2cam': GOLD['in_%d_Tr_velo2cam' % i]})
              for i in range(N_FRAMES)]
    batch = {'frame_id': [str(GOLD['in_%d_frame_id' % i]) for i in range(N_FRAMES)], 'calib': calibs,
             'image_shape': torch.from_numpy(np.stack([GOLD['in_%d_image_shape' % i] for i in range(N_FRAMES)])).cuda()}
    preds = [{'pred_boxes': torch.from_numpy(GOLD['in_%d_boxes' % i][:, :ncol].copy()).cuda(),
              'pred_scores': torch.from_numpy(GOLD['in_%d_scores' % i]).cuda(),
              'pred_labels': torch.from_numpy(GOLD['in_%d_labels' % i]).cuda()} for i in range(N_FRAMES)]
    return batch, preds


def test_kernel_equals_c_oracle(oracle_ops):
    from de6d_amd.ops import fused
    batch, preds = batch_and_preds(9)
    boxes = torch.cat([p['pred_boxes'] for p in preds])
    counts = [len(p['pred_scores']) for p in preds]
    scene_of = np.repeat(np.arange(N_FRAMES, dtype=np.int32), counts)
    calib = np.stack([c.packed(s) for c, s in zip(batch['calib'], batch['image_shape'].cpu().numpy())])
    got = fused.kitti_annos(boxes, torch.from_numpy(scene_of).cuda(), torch.from_numpy(calib).cuda()).cpu().numpy()
    ref = oracle_ops.kitti_annos(boxes.cpu().numpy(), scene_of, calib)
    assert np.array_equal(got, ref)
    # a big random batch too (the launch spans several workgroups)
    rng = np.random.default_rng(3)
    big = np.tile(boxes.cpu().numpy(), (40, 1)) + rng.normal(0, 0.3, (40 * len(scene_of), 9)).astype(np.float32)
    big_scene = rng.integers(0, N_FRAMES, len(big)).astype(np.int32)
    got = fused.kitti_annos(torch.from_numpy(big).cuda(), torch.from_numpy(big_scene).cuda(), torch.from_numpy(calib).cuda())
    assert np.array_equal(got.cpu().numpy(), oracle_ops.kitti_annos(big, big_scene, calib))


def parse_line(line):
    head, *nums = line.split(' ')
    return head, np.array([float(v) for v in nums])


@pytest.mark.parametrize('tag,ncol', [('kitti', 7), ('sloped', 9)])
def test_prediction_dicts_match_reference(tmp_path, tag, ncol):
    from de6d_amd.pcdet import datasets
    cls = datasets.KittiDataset if tag == 'kitti' else datasets.SlopedKittiDataset
    batch, preds = batch_and_preds(ncol)
    annos = cls.generate_prediction_dicts(batch, preds, NAMES, output_path=Path(tmp_path))
    assert len(annos) == N_FRAMES
    for i, a in enumerate(annos):
        assert a['frame_id'] == batch['frame_id'][i]
        assert [str(n) for n in a['name']] == [str(n) for n in GOLD['%s_%d_name' % (tag, i)]]
        for key, atol in (('location', TOL), ('dimensions', 0.0), ('rotation_y', 1e-6), ('alpha', TOL), ('bbox', 5e-3),
                          ('score', 0.0), ('boxes_lidar', 0.0)) + ((('pitch', 0.0), ('roll', 0.0)) if tag == 'sloped' else ()):
            want = GOLD['%s_%d_%s' % (tag, i, key)]
            assert np.asarray(a[key]).shape == want.shape, (key, i)
            assert np.allclose(np.asarray(a[key], np.float64), want, atol=atol, rtol=1e-5 if key == 'bbox' else 0), (key, i)
        want_lines = str(GOLD['%s_%d_txt' % (tag, i)]).splitlines()
        got_lines = (tmp_path / ('%s.txt' % a['frame_id'])).read_text().splitlines()
        assert len(got_lines) == len(want_lines)
        for g, w in zip(got_lines, want_lines):
            (gn, gv), (wn, wv) = parse_line(g), parse_line(w)
            assert gn == wn and gv.shape == wv.shape and np.allclose(gv, wv, atol=6e-3)   # 4-decimal text, pixel columns


class _Frames(torch.utils.data.Dataset):
    """synthetic stand-in for a KITTI split: frames, calibration, image shape, ids"""
    class_names = ['Car']

    def __init__(self, n_frames, n_points):
        from tests.util import make_scene
        from de6d_amd.pcdet.utils.calibration_kitti import Calibration
        self.scenes = [make_scene(900 + i, n_points) for i in range(n_frames)]
        self.calib = Calibration({'P2': GOLD['in_0_P2'], 'R0': GOLD['in_0_R0'], 'Tr_velo2cam': GOLD['in_0_Tr_velo2cam']})

    def __len__(self):
        return len(self.scenes)

    def __getitem__(self, i):
        return {'points': self.scenes[i], 'frame_id': '%06d' % i, 'calib': self.calib, 'image_shape': np.array([375, 1242], np.int32)}

    @staticmethod
    def collate_batch(items):
        pts = np.concatenate([np.pad(d['points'], ((0, 0), (1, 0)), constant_values=i) for i, d in enumerate(items)], 0)
        return {'points': pts.astype(np.float32), 'frame_id': [d['frame_id'] for d in items], 'calib': [d['calib'] for d in items],
                'image_shape': np.stack([d['image_shape'] for d in items]), 'batch_size': len(items)}

    generate_prediction_dicts = None
    def evaluation(self, det_annos, class_names, **kwargs):
        raise NotImplementedError('no ground truth in the synthetic split')


def test_eval_one_epoch_end_to_end(tmp_path):
    from de6d_amd.pcdet.datasets import SlopedKittiDataset
    from de6d_amd.runtime import load_config, build_model
    from de6d_amd.tools.eval_utils.eval_utils import eval_one_epoch
    cfg = load_config('synthetic_models/det6d_tiny.yaml')
    cfg.LOCAL_RANK = 0
    model = build_model(cfg, seed=21, device='cuda')
    ds = _Frames(5, 2048)
    ds.generate_prediction_dicts = SlopedKittiDataset.generate_prediction_dicts
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, collate_fn=_Frames.collate_batch)
    log = logging.getLogger('eval_test')
    ret = eval_one_epoch(cfg, model, loader, 'test', log, dist_test=False, save_to_file=True, result_dir=Path(tmp_path))
    assert set(ret) == {'recall/%s_%s' % (s, t) for s in ('roi', 'rcnn') for t in cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST}
    annos = pickle.load(open(tmp_path / 'result.pkl', 'rb'))
    assert [a['frame_id'] for a in annos] == ['%06d' % i for i in range(5)]
    for a in annos:
        k = len(a['name'])
        assert a['bbox'].shape == (k, 4) and a['boxes_lidar'].shape[0] == k and len(a['pitch']) == k
        lines = (tmp_path / 'final_result' / 'data' / ('%s.txt' % a['frame_id'])).read_text().splitlines()
        assert len(lines) == k and all(len(ln.split(' ')) == 18 for ln in lines)
    assert sum(len(a['name']) for a in annos) > 0


def test_padded_result_block_fast_path_equals_the_generic_path():
    """captured passes hand out per-frame views of one padded (B, P, C) block: convert_batch converts the whole block with
    one launch (no per-frame concatenation); a sub-range of the block's frames (a step of a coalesced pass), frames with no
    detections, and tensors that are NOT such views (generic path) must all give the same annotation dictionaries"""
    from de6d_amd.pcdet import datasets
    from de6d_amd.pcdet.datasets.kitti import kitti_dataset as kd
    from de6d_amd.pcdet.utils.calibration_kitti import Calibration
    rng = np.random.default_rng(9)
    nb, pmax, ncol = 6, 100, 9
    calibs = [Calibration({'P2': GOLD['in_%d_P2' % (i % N_FRAMES)], 'R0': GOLD['in_%d_R0' % (i % N_FRAMES)],
                           'Tr_velo2cam': GOLD['in_%d_Tr_velo2cam' % (i % N_FRAMES)]}) for i in range(nb)]
    boxes = torch.from_numpy((rng.normal(size=(nb, pmax, ncol)) * [10, 5, 1, 1, 1, 1, 1, 0.1, 0.1] + [25, 0, -1, 4, 2, 1.5, 0, 0, 0]).astype(np.float32)).cuda()
    scores = torch.from_numpy(rng.uniform(0.1, 1, (nb, pmax)).astype(np.float32)).cuda()
    labels = torch.from_numpy(rng.integers(1, 4, (nb, pmax))).cuda()
    counts = [100, 0, 37, 1, 64, 5]
    views = [{'pred_boxes': boxes[i, :k], 'pred_scores': scores[i, :k], 'pred_labels': labels[i, :k]} for i, k in enumerate(counts)]
    copies = [{k_: v.clone() for k_, v in p.items()} for p in views]
    for lo, hi in ((0, 6), (2, 5), (1, 2), (3, 4), (5, 6)):     # the last two: P > 4 x the largest count -> the generic path serves views too
        batch = {'frame_id': ['%06d' % i for i in range(lo, hi)], 'calib': calibs[lo:hi],
                 'image_shape': np.tile(np.array([[375, 1242]], np.int32), (hi - lo, 1))}
        assert kd._padded_block(views[lo:hi]) is not None
        assert kd._padded_block(copies[lo:hi]) is None
        fast = datasets.SlopedKittiDataset.generate_prediction_dicts(batch, views[lo:hi], NAMES)
        slow = datasets.SlopedKittiDataset.generate_prediction_dicts(batch, copies[lo:hi], NAMES)
        assert len(fast) == len(slow) == hi - lo
        for a, b in zip(fast, slow):
            assert a.keys() == b.keys()
            for key in a:
                assert np.array_equal(a[key], b[key]), key
