"""Reference-shaped neighbour queries at every nsample / size the reference accepts (ball_query_gpu.cu:15-130 has no cap): the
grid-hashed kernel serves nsample <= 64 on n >= 2048, everything else must fall back to the brute-force kernels — never raise
(round-3 regression: n >= 2048 with 64 < nsample <= 128 was routed to the grid kernel, which rejects it)."""
import numpy as np
import pytest
import torch

from tests.util import make_batch, beam_batch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def pn():
    from de6d_amd.ops import pointnet2_batch_hip
    return pointnet2_batch_hip


def _check_all_three(pn, oracle_ops, xyz, new_xyz, r_in, r_out, ns):
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    x, q = dev(xyz), dev(new_xyz)
    cnt = torch.zeros((b, m), dtype=torch.int32, device="cuda")
    idx = torch.zeros((b, m, ns), dtype=torch.int32, device="cuda")
    pn.ball_query_dilated_wrapper(b, n, m, r_in, r_out, ns, q, x, cnt, idx)
    ocnt, oidx = oracle_ops.ball_query_dilated(r_in, r_out, ns, xyz, new_xyz)
    np.testing.assert_array_equal(cnt.cpu().numpy(), ocnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), oidx)
    cnt.zero_(); idx.zero_()
    pn.ball_query_cnt_wrapper(b, n, m, r_out, ns, q, x, cnt, idx)
    ocnt, oidx = oracle_ops.ball_query_cnt(r_out, ns, xyz, new_xyz)
    np.testing.assert_array_equal(cnt.cpu().numpy(), ocnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), oidx)
    idx.zero_()
    pn.ball_query_wrapper(b, n, m, r_out, ns, q, x, idx)
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle_ops.ball_query(r_out, ns, xyz, new_xyz))


@pytest.mark.parametrize("n", [2048, 16384])
@pytest.mark.parametrize("ns", [64, 65, 96, 128])
def test_large_nsample_on_large_clouds(pn, oracle_ops, n, ns):
    b, m = 2, 96
    xyz = beam_batch(11, b, n)[..., :3]                  # ray-cast density: balls near the sensor overflow every nsample
    new_xyz = np.ascontiguousarray(xyz[:, ::n // m][:, :m] + np.float32(0.01))
    new_xyz[:, 0] = 1000.0                                # a centre with no neighbour at all
    _check_all_three(pn, oracle_ops, xyz, new_xyz, 0.4, 2.5, ns)


def test_grid_supported_is_the_routing_predicate(pn):
    from de6d_amd import _lib as L
    sup = L.lib().det6d_ball_query_grid_supported
    assert sup(16384, 64, 64) == 1 and sup(16384, 65, 1) == 0 and sup(16384, 1, 65) == 0
    assert sup(98304, 16, 32) == 1 and sup(98305, 16, 32) == 0 and sup(0, 16, 32) == 0
    assert pn._grid_takes(2048, 64) and not pn._grid_takes(2048, 65) and not pn._grid_takes(2047, 16)


def test_random_reference_shaped_queries(pn, oracle_ops):
    """seeded sweep over (b, n, m, nsample, r_in, r_out) of the three reference entries vs the oracle"""
    rng = np.random.default_rng(2024)
    for it in range(24):
        b = int(rng.integers(1, 4))
        n = int(rng.choice([1, 7, 63, 64, 65, 500, 2047, 2048, 2049, 5000, 16384]))
        m = int(rng.integers(1, 200))
        ns = int(rng.choice([1, 2, 3, 8, 15, 16, 31, 32, 33, 48, 64, 65, 100, 128, 200]))
        r_out = float(rng.choice([0.05, 0.3, 1.0, 3.0, 50.0]))
        r_in = float(rng.choice([0.0, 0.0, r_out * 0.5, r_out]))
        xyz = (beam_batch if it % 2 else make_batch)(100 + it, b, max(n, 64))[:, :n, :3]
        xyz = np.ascontiguousarray(xyz)
        pick = rng.integers(0, n, (b, m))
        new_xyz = np.stack([xyz[i, pick[i]] for i in range(b)]) + rng.normal(0, 0.05, (b, m, 3)).astype(np.float32)
        _check_all_three(pn, oracle_ops, xyz, np.ascontiguousarray(new_xyz.astype(np.float32)), r_in, r_out, ns)


@pytest.mark.parametrize("ns,ncols,ldx", [(12, 40, 40), (5, 33, 36), (1, 8, 8), (100, 64, 68), (24, 128, 128)])
def test_group_maxpool_any_nsample(oracle_ops, ns, ncols, ldx):
    from de6d_amd.ops import fused
    rng = np.random.default_rng(ns)
    groups = 77
    x = np.maximum(rng.normal(size=(groups * ns, ldx)), 0).astype(np.float32)
    cnt = rng.integers(0, 3, groups).astype(np.int32)
    out = torch.full((groups, ncols + 8), -7.0, device="cuda")
    fused.group_maxpool(dev(x), ns, ncols, dev(cnt), out, 4)
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got[:, 4:4 + ncols], oracle_ops.group_maxpool(x, ns, ncols, cnt))
    assert (got[:, :4] == -7).all() and (got[:, 4 + ncols:] == -7).all()     # nothing outside the slice is touched


def test_whole_model_with_any_nsample(oracle_ops):
    """a Det6D whose NSAMPLE values are not the ones the fused pooling epilogues know (8 / 16 / 32): every SA layer and the
    head fall back to a stored last layer + det6d_group_maxpool instead of raising (pointnet2_modules.py:465-472 accepts any
    nsample); still the oracle's results bit for bit"""
    from de6d_amd.runtime import load_config, build_model
    from oracle import model as omodel
    cfg = load_config('synthetic_models/det6d_tiny.yaml')
    cfg.MODEL.BACKBONE_3D.SA_CONFIG.NSAMPLE = [[12, 24], [16, 20], [6, 40]]
    cfg.MODEL.POINT_HEAD.SA_CONFIG.NSAMPLE = [10, 48]
    model = build_model(cfg, seed=11, device='cuda:0')
    b, n = 2, 2048
    batch = make_batch(78, b, n)
    pts = np.concatenate([np.repeat(np.arange(b, dtype=np.float32), n)[:, None], batch.reshape(b * n, 4)], 1).astype(np.float32)
    bd = {'batch_size': b, 'points': torch.from_numpy(pts).cuda()}
    with torch.no_grad():
        pred, _ = model(bd)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ref = omodel.forward(cfg.MODEL, sd, pts, b)
    np.testing.assert_array_equal(bd['batch_box_preds'].cpu().numpy(), ref['batch_box_preds'])
    np.testing.assert_array_equal(bd['batch_cls_preds'].cpu().numpy(), ref['batch_cls_preds'])
    for got, want in zip(pred, ref['pred_dicts']):
        np.testing.assert_array_equal(got['pred_boxes'].cpu().numpy(), want['pred_boxes'])
        np.testing.assert_array_equal(got['pred_scores'].cpu().numpy(), want['pred_scores'])


@pytest.mark.parametrize("scene", ["uniform", "beam", "blob", "dup"])
@pytest.mark.parametrize("smin,split", [(1, 1), (4, 4), (2, 4)])
def test_engine_query_counts_parts_and_feeds_the_placement_only_builder(oracle_ops, scene, smin, split):
    """det6d_ball_query_pair_grid_lists + det6d_compact_groups_pair_counted (round 5): counts and the index slots the list
    builder reads equal the oracle's query; the per-block part tables the query leaves in hdr equal a count over the oracle's
    counts; the placement-only builder fed by them gives the lists of det6d_compact_groups_pair (count + place) word for word.
    Light centres (a lane walks its candidates), heavy centres (dense blob / ray-cast near field: a wave each) and both kinds
    inside one wave occur."""
    from de6d_amd.ops import fused
    from tests.util import beam_batch, make_batch
    rng = np.random.default_rng(77)
    b, n, m = 3, 16384, 1024
    if scene == "uniform":
        xyz = make_batch(910, b, n)[..., :3]
    elif scene == "beam":
        xyz = beam_batch(911, b, n)[..., :3]
    elif scene == "dup":
        xyz = make_batch(912, b, n, dup_frac=0.3)[..., :3]
    else:
        xyz = (rng.normal(size=(b, n, 3)) * [6.0, 6.0, 0.4]).astype(np.float32)
        xyz[:, ::2] *= np.float32(8.0)                       # a dense core inside a sparse halo
    xyz = np.ascontiguousarray(xyz, np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, rng.choice(n, m, replace=False)] + np.float32(0.004))
    new_xyz[:, 5] = 900.0                                    # empty balls
    sa, sb = (0.0, 0.2, 16), (0.2, 0.8, 32)
    old = fused.COMPACT_SMIN, fused.COMPACT_SPLIT
    fused.COMPACT_SMIN, fused.COMPACT_SPLIT = smin, split
    try:
        got = fused.ball_query_pair_lists(dev(xyz), dev(new_xyz), sa, sb)
        assert got is not None
        ca, ia, cb, ib, la, lb = got
        oca, oia = oracle_ops.ball_query_dilated(sa[0], sa[1], sa[2], xyz, new_xyz)
        ocb, oib = oracle_ops.ball_query_dilated(sb[0], sb[1], sb[2], xyz, new_xyz)
        for c, i, oc, oi, ns, lst in ((ca, ia, oca, oia, 16, la), (cb, ib, ocb, oib, 32, lb)):
            c, i = c.cpu().numpy(), i.cpu().numpy()
            np.testing.assert_array_equal(c, oc)
            need = np.maximum(4, 2 ** np.ceil(np.log2(np.maximum(oc, 1))).astype(np.int64))
            live = np.arange(ns)[None, None, :] < need[..., None]
            np.testing.assert_array_equal(np.where(live, i, 0), np.where(live, oi, 0))
            # the part tables: 7 ints per block of 256 centres
            sm, sp = min(smin, ns), min(max(fused.COMPACT_SPLIT, min(smin, ns)), ns)
            kk = np.clip(oc.reshape(-1), 1, ns)
            pow2 = np.maximum(sm, 2 ** np.ceil(np.log2(kk)).astype(np.int64))
            rows = np.where(kk > sp, (kk + sp - 1) // sp * sp, pow2)
            table = lst.hdr.cpu().numpy()[16:16 + 7 * (b * m // 256)].reshape(-1, 7)
            for cls in range(6):
                np.testing.assert_array_equal(table[:, cls], ((rows & (32 >> cls)) != 0).reshape(-1, 256).sum(1))
            np.testing.assert_array_equal(table[:, 6], np.minimum(oc, ns).reshape(-1, 256).sum(1))
        if scene in ("beam", "blob"):
            assert (ocb == 32).any()                         # capped balls: the wave-per-centre route with its pruning threshold
        pooled = torch.full((b * m, 64), 7.0, device='cuda')
        counted = fused.compact_groups_pair([(ca, ia), (cb, ib)], n, pooled, [(0, 32), (32, 32)], counted=(la, lb))
        # reference lists: the padded query + the counting builder
        fa, fia, fb, fib = fused.ball_query_pair(dev(xyz), dev(new_xyz), sa, sb, grid=True)
        pooled2 = torch.full((b * m, 64), 7.0, device='cuda')
        plain = fused.compact_groups_pair([(fa, fia), (fb, fib)], n, pooled2, [(0, 32), (32, 32)])
        for x, y in zip(counted, plain):
            hx, hy = x.hdr.cpu().numpy(), y.hdr.cpu().numpy()
            np.testing.assert_array_equal(hx[:10], hy[:10])
            live_rows = int(hx[0])
            np.testing.assert_array_equal(x.crow_p.cpu().numpy()[:live_rows], y.crow_p.cpu().numpy()[:live_rows])
            np.testing.assert_array_equal(x.crow_c.cpu().numpy()[:live_rows], y.crow_c.cpu().numpy()[:live_rows])
        assert torch.equal(pooled, pooled2)
    finally:
        fused.COMPACT_SMIN, fused.COMPACT_SPLIT = old
