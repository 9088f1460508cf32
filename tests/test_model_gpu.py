"""Whole-pipeline GPU parity: the HIP Det6D (de6d_amd.pcdet) against the CPU oracle model
(oracle/model.py) on the same seeded scenes and weights.  Everything is compared BIT-EXACT:
sampled indices, ball-query counts, layer features, scores, votes, logits, decoded boxes and
the final detections (keep order included)."""
import numpy as np
import pytest
import torch

from tests.util import make_batch

pytestmark = pytest.mark.gpu


def flat_points(batch):
    b, n, _ = batch.shape
    bidx = np.repeat(np.arange(b, dtype=np.float32), n)[:, None]
    return np.concatenate([bidx, batch.reshape(b * n, 4)], 1).astype(np.float32)


def run_both(cfg_name, b, n, seed, tilt=False):
    from de6d_amd.runtime import load_config, build_model
    from oracle import model as omodel
    cfg = load_config(cfg_name)
    model = build_model(cfg, seed=seed, device='cuda')
    pts = flat_points(make_batch(seed, b, n, tilt=tilt))
    bd = {'batch_size': b, 'points': torch.from_numpy(pts).cuda()}
    with torch.no_grad():
        pred, _ = model(bd)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ref = omodel.forward(cfg.MODEL, sd, pts, b)
    return cfg, bd, pred, ref


def check(bd, pred, ref, b):
    for lvl, xyz in enumerate(ref['l_xyz']):
        got = bd['point_coords_list'][lvl].cpu().numpy()
        np.testing.assert_array_equal(got[:, 1:], xyz.reshape(-1, 3), err_msg='sampled xyz level %d' % lvl)
        np.testing.assert_array_equal(got[:, 0], np.repeat(np.arange(b), xyz.shape[1]))
        s = ref['l_scores'][lvl]
        if s is None:
            assert bd['point_scores_list'][lvl] is None
        else:
            np.testing.assert_array_equal(bd['point_scores_list'][lvl].cpu().numpy().reshape(s.shape), s,
                                          err_msg='confidence scores level %d' % lvl)
    np.testing.assert_array_equal(bd['point_features'].cpu().numpy(), ref['point_features'])
    np.testing.assert_array_equal(bd['point_vote_coords'].cpu().numpy()[:, 1:], ref['point_vote_coords'])
    np.testing.assert_array_equal(bd['point_candidate_coords'].cpu().numpy()[:, 1:], ref['point_candidate_coords'])
    np.testing.assert_array_equal(bd['batch_cls_preds'].cpu().numpy(), ref['batch_cls_preds'])
    np.testing.assert_array_equal(bd['point_reg_preds'].cpu().numpy(), ref['point_reg_preds'])
    np.testing.assert_array_equal(bd['batch_box_preds'].cpu().numpy(), ref['batch_box_preds'])
    for got, want in zip(pred, ref['pred_dicts']):
        np.testing.assert_array_equal(got['pred_boxes'].cpu().numpy(), want['pred_boxes'])
        np.testing.assert_array_equal(got['pred_scores'].cpu().numpy(), want['pred_scores'])
        np.testing.assert_array_equal(got['pred_labels'].cpu().numpy(), want['pred_labels'])


@pytest.mark.parametrize("seed,tilt", [(11, False), (12, True), (13, False)])
def test_tiny_model_bit_exact(oracle_ops, seed, tilt):
    cfg, bd, pred, ref = run_both('synthetic_models/det6d_tiny.yaml', 3, 2048, seed, tilt)
    check(bd, pred, ref, 3)
    assert sum(len(p['pred_scores']) for p in pred) > 0


def test_full_car_model_bit_exact(oracle_ops):
    """BASELINE config 2 shapes (16384-point scenes, full-width network), 2 scenes"""
    cfg, bd, pred, ref = run_both('kitti_models/det6d_car.yaml', 2, 16384, 21)
    check(bd, pred, ref, 2)
    assert bd['batch_box_preds'].shape == (2 * 256, 9)


def test_sloped_scene_pitch_branch(oracle_ops):
    """BASELINE config 3: tilted scenes through the ground-aware decoder; the pitch != 0 branch must
    occur (the pitch == 0 branch is covered by test_ops_gpu.py::test_head_elementwise_bit_exact)"""
    cfg, bd, pred, ref = run_both('slopedkitti_models/det6d_car.yaml', 1, 16384, 31, tilt=True)
    check(bd, pred, ref, 1)
    pitch = ref['batch_box_preds'][:, 7]
    assert (pitch != 0).any()


def test_three_class_model(oracle_ops):
    """BASELINE config 4 head shape (cls out-channels 3)"""
    cfg, bd, pred, ref = run_both('kitti_models/det6d_3class.yaml', 1, 16384, 41)
    check(bd, pred, ref, 1)
    assert bd['batch_cls_preds'].shape[1] == 3
    labels = np.concatenate([p['pred_labels'].cpu().numpy() for p in pred])
    assert labels.min() >= 1 and labels.max() <= 3


def test_reference_shaped_sa_forward_matches_rows_path(oracle_ops):
    """the channel-major forward(xyz, features) wrapper and forward_rows() agree"""
    from de6d_amd.runtime import load_config, build_model
    cfg = load_config('synthetic_models/det6d_tiny.yaml')
    model = build_model(cfg, seed=5, device='cuda')
    sa = model.backbone_3d.SA_modules[0]
    pts = make_batch(5, 2, 2048)
    xyz = torch.from_numpy(np.ascontiguousarray(pts[..., :3])).cuda()
    feats = torch.from_numpy(np.ascontiguousarray(pts[..., 3:].transpose(0, 2, 1))).cuda()
    with torch.no_grad():
        nx, nf, ns = sa(xyz, feats)
    assert nx.shape == (2, 512, 3) and nf.shape == (2, 16, 512) and ns.shape == (2, 512)
    from oracle import model as omodel
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    spec = omodel.backbone_specs(cfg.MODEL)[0]
    rx, rf, rs, _ = omodel.sa_layer(sd, 'backbone_3d.SA_modules.0', spec, pts[..., :3].copy(), feats.cpu().numpy())
    np.testing.assert_array_equal(nx.cpu().numpy(), rx)
    np.testing.assert_array_equal(nf.cpu().numpy(), rf)
    np.testing.assert_array_equal(ns.cpu().numpy(), rs)


def test_waymo_scale_65536_points(oracle_ops):
    """BASELINE config 5 shapes: 65536-point scene, every per-layer point count x4 (the FPS of the
    first layer takes the memory-resident generic kernel: N is beyond the register-resident sizes)"""
    cfg, bd, pred, ref = run_both('synthetic_models/det6d_65536.yaml', 1, 65536, 51)
    check(bd, pred, ref, 1)
    assert bd['point_coords_list'][0].shape[0] == 16384 and bd['batch_box_preds'].shape == (1024, 9)


def test_ragged_and_tiny_inputs(oracle_ops):
    """edge cases of the op API: N not a power of two, N < 64, M == N, single point"""
    from de6d_amd.ops import pointnet2_batch_hip as pn
    for n, m in [(5000, 128), (333, 333), (63, 10), (2, 2)]:
        xyz = make_batch(60 + n, 2, max(n, 8))[:, :n, :3].copy()
        x = torch.from_numpy(xyz).cuda()
        temp = torch.full((2, n), 1e10, device='cuda')
        idx = torch.zeros((2, m), dtype=torch.int32, device='cuda')
        pn.farthest_point_sampling_wrapper(2, n, m, x, temp, idx)
        np.testing.assert_array_equal(idx.cpu().numpy(), oracle_ops.fps(xyz, m))
    # empty batch / zero samples are no-ops
    e = torch.zeros((0, 8, 3), device='cuda')
    pn.farthest_point_sampling_wrapper(0, 8, 4, e, torch.zeros((0, 8), device='cuda'), torch.zeros((0, 4), dtype=torch.int32, device='cuda'))


def test_graph_replay_equals_eager(oracle_ops):
    """the hipGraph runner used by bench.py returns exactly the eager detections, replay after replay"""
    from de6d_amd.runtime import load_config, build_model, GraphedDet6D
    cfg = load_config('synthetic_models/det6d_tiny.yaml')
    model = build_model(cfg, seed=9, device='cuda')
    b, n = 2, 2048
    pts = torch.from_numpy(flat_points(make_batch(90, b, n))).cuda()
    with torch.no_grad():
        eager, _ = model({'batch_size': b, 'points': pts})
    runner = GraphedDet6D(model, b, n)
    for _ in range(3):
        got = runner.launch(pts).finalize()
        for g, e in zip(got, eager):
            assert torch.equal(g['pred_boxes'], e['pred_boxes']) and torch.equal(g['pred_scores'], e['pred_scores'])
    pts2 = torch.from_numpy(flat_points(make_batch(91, b, n))).cuda()
    with torch.no_grad():
        eager2, _ = model({'batch_size': b, 'points': pts2})
    got2 = runner.launch(pts2).finalize()
    for g, e in zip(got2, eager2):
        assert torch.equal(g['pred_boxes'], e['pred_boxes'])


def test_pass_group_equals_eager(oracle_ops):
    """Det6DGroup (bench.py's runner: graph segments per pass, the first sampler of the group's passes as one
    high-priority launch) returns exactly the eager detections of every pass, with different inputs per pass and
    with partially filled groups"""
    from de6d_amd.runtime import load_config, build_model, Det6DGroup
    cfg = load_config('synthetic_models/det6d_tiny.yaml')
    model = build_model(cfg, seed=9, device='cuda')
    b, n, k = 2, 2048, 3
    group = Det6DGroup(model, b, n, k, torch.cuda.Stream(priority=-1))
    batches = [torch.from_numpy(flat_points(make_batch(300 + 7 * j, b, n))).cuda() for j in range(k)]
    eager = []
    with torch.no_grad():
        for pts in batches:
            eager.append(model({'batch_size': b, 'points': pts})[0])
    for rep in range(3):
        count = k if rep != 1 else 2
        for r, pts in zip(group.runners[:count], batches):
            r.points.copy_(pts)
        torch.cuda.synchronize()
        passes = group.launch(count=count)
        assert len(passes) == count
        for r, want in zip(passes, eager):
            for g, e in zip(r.finalize(), want):
                assert torch.equal(g['pred_boxes'], e['pred_boxes']) and torch.equal(g['pred_scores'], e['pred_scores'])
                assert torch.equal(g['pred_labels'], e['pred_labels'])


def test_two_stage_launch_with_producer_and_lazy_coords(oracle_ops):
    """launch_front() / launch_rest() issued separately with an input PRODUCER (a callable that fills pass.points on the
    sampler stream, as bench.py's raw-frames leg does), several fronts ahead of the rests; and the lazily built
    point_coords_list of a captured pass equals the eager one"""
    from de6d_amd.runtime import load_config, build_model, Det6DGroup
    cfg = load_config('synthetic_models/det6d_tiny.yaml')
    model = build_model(cfg, seed=5, device='cuda')
    b, n, k = 2, 2048, 2
    samp = torch.cuda.Stream()
    mains = [torch.cuda.Stream() for _ in range(2)]
    groups = [Det6DGroup(model, b, n, k, samp, main_streams=mains) for _ in range(3)]
    batches = [torch.from_numpy(flat_points(make_batch(500 + j, b, n))).cuda() for j in range(6)]
    eager, coords = [], []
    with torch.no_grad():
        for pts in batches:
            bd = {'batch_size': b, 'points': pts}
            eager.append(model(bd)[0])
            coords.append([c.clone() for c in bd['point_coords_list']])
    torch.cuda.synchronize()
    feed = iter(range(6))

    def produce(r):
        r.points.copy_(batches[next(feed)], non_blocking=True)
    for g in groups:                      # all three fronts first (prefetch), then the rests
        g.launch_front(produce)
    j = 0
    for g in groups:
        for r in g.launch_rest():
            for got, want in zip(r.finalize(), eager[j]):
                assert torch.equal(got['pred_boxes'], want['pred_boxes']) and torch.equal(got['pred_scores'], want['pred_scores'])
            lazy = r.batch_dict['point_coords_list']
            assert len(lazy) == len(coords[j]) and isinstance(lazy, list)
            for lvl in range(len(lazy)):
                assert torch.equal(lazy[lvl], coords[j][lvl])
            assert torch.equal(lazy[-1], coords[j][-1]) and torch.equal(lazy[0:2][1], coords[j][1])
            j += 1
    assert j == 6


def test_generic_post_processing_route_matches_fused(oracle_ops):
    """class_agnostic_nms + nms_gpu through the op-level API (the route taken for non-fusable configs)
    selects the same boxes as the fused post-processing kernels and the oracle"""
    from de6d_amd.pcdet.config import EasyDict
    from de6d_amd.pcdet.models.model_utils.model_nms_utils import class_agnostic_nms
    from de6d_amd.ops import fused
    from tests.util import random_boxes
    rng = np.random.default_rng(4)
    p = 300
    boxes = np.zeros((p, 9), np.float32)
    boxes[:, :7] = random_boxes(8, p, spread=25.0)
    logits = (rng.normal(size=(p, 1)) * 2).astype(np.float32)
    logits[11] = logits[5]
    nms_cfg = EasyDict(NMS_TYPE='nms_gpu', NMS_THRESH=0.1, NMS_PRE_MAXSIZE=200, NMS_POST_MAXSIZE=50, MULTI_CLASSES_NMS=False)
    b, s = torch.from_numpy(boxes).cuda(), fused.sigmoid_pow(torch.from_numpy(logits[:, 0].copy()).cuda(), 1.0)
    sel, sel_scores = class_agnostic_nms(s, b, nms_cfg, score_thresh=0.3)
    ob, osc, ol, oi, oc = oracle_ops.postprocess(logits, boxes, 1, 0.3, 200, 50, 0.1)
    np.testing.assert_array_equal(sel.cpu().numpy(), oi[0, :oc[0]])
    np.testing.assert_array_equal(sel_scores.cpu().numpy(), osc[0, :oc[0]])
    fb, fs, fl, fi, fc = fused.postprocess(torch.from_numpy(logits).cuda(), b, 1, 0.3, 200, 50, 0.1)
    np.testing.assert_array_equal(fi.cpu().numpy(), oi)
    # nothing above threshold
    sel, _ = class_agnostic_nms(s, b, nms_cfg, score_thresh=2.0)
    assert sel.numel() == 0


def test_determinism_and_batch_shapes(oracle_ops):
    """same input -> identical output run after run and stream after stream; odd batch sizes work"""
    from de6d_amd.runtime import load_config, build_model
    cfg = load_config('synthetic_models/det6d_tiny.yaml')
    model = build_model(cfg, seed=17, device='cuda')
    for b in (1, 3, 5):
        pts = torch.from_numpy(flat_points(make_batch(200 + b, b, 2048))).cuda()
        outs = []
        for rep in range(3):
            stream = torch.cuda.Stream() if rep else torch.cuda.current_stream()
            with torch.no_grad(), torch.cuda.stream(stream):
                pred, _ = model({'batch_size': b, 'points': pts})
            stream.synchronize()
            outs.append(pred)
        for rep in (1, 2):
            for p0, p1 in zip(outs[0], outs[rep]):
                assert torch.equal(p0['pred_boxes'], p1['pred_boxes']) and torch.equal(p0['pred_scores'], p1['pred_scores'])
        # a scene's detections do not depend on what else is in the batch
        with torch.no_grad():
            single, _ = model({'batch_size': 1, 'points': pts[:2048].clone()})
        assert torch.equal(single[0]['pred_boxes'], outs[0][0]['pred_boxes'])


def test_degenerate_scenes_through_the_whole_model(oracle_ops):
    """scenes the reference's `sample_points` padding and sparse frames produce in the extreme: every point identical (all
    distances 0: every FPS round is a tie, every ball is full of one point), two distinct points, an exact lattice (ties in every
    sampler and in every ball-query distance test), a dense blob (every ball capped at nsample, hundreds of hits per shell).
    Tiny widths, 2048 points, through the eager model AND the full-width model on 16384 points for the blob; bit-exact."""
    from de6d_amd.runtime import load_config, build_model
    from oracle import model as omodel
    rng = np.random.default_rng(3)
    n = 2048
    same = np.tile(np.array([[12.0, 3.0, -1.0, 0.5]], np.float32), (n, 1))
    two = same.copy(); two[1::2] = [12.4, 3.2, -0.9, 0.1]
    g = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(8), indexing='ij'), -1).reshape(-1, 3)[:n]
    lattice = np.concatenate([g * np.float32(0.25) + np.float32([5, -2, -2]), np.full((n, 1), 0.3)], 1).astype(np.float32)
    blob = np.concatenate([rng.normal(size=(n, 3)) * [1.0, 1.0, 0.2] + [20, 0, -1], rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
    cfg = load_config('synthetic_models/det6d_tiny.yaml')
    model = build_model(cfg, seed=17, device='cuda')
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    batch = np.stack([same, two, lattice, blob], 0)
    pts = flat_points(batch)
    bd = {'batch_size': 4, 'points': torch.from_numpy(pts).cuda()}
    with torch.no_grad():
        pred, _ = model(bd)
    check(bd, pred, omodel.forward(cfg.MODEL, sd, pts, 4), 4)
    # full width, 16384 points: a dense blob beside a normal scene (every SA1 ball of the blob is capped)
    cfg = load_config('kitti_models/det6d_car.yaml')
    model = build_model(cfg, seed=18, device='cuda')
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    n = 16384
    blob = np.concatenate([rng.normal(size=(n, 3)) * [2.0, 2.0, 0.3] + [20, 0, -1], rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
    batch = np.stack([blob, make_batch(77, 1, n)[0]], 0)
    pts = flat_points(batch)
    bd = {'batch_size': 2, 'points': torch.from_numpy(pts).cuda()}
    with torch.no_grad():
        pred, _ = model(bd)
    check(bd, pred, omodel.forward(cfg.MODEL, sd, pts, 2), 2)
