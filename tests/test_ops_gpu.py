"""GPU parity tests: every C-ABI op of libdet6d_hip.so (through the reference-shaped module
functions) against the CPU oracle on the same seeded inputs.  Bit-exact for indices, masks and
for everything computed with the shared deterministic arithmetic."""
import os

import numpy as np
import pytest
import torch

from tests.util import make_batch, random_boxes

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ext():
    from de6d_amd.ops import pointnet2_batch_hip, iou3d_nms_hip, fused
    return pointnet2_batch_hip, iou3d_nms_hip, fused


def hip_fps(pn, xyz, m, weights=None):
    b, n, _ = xyz.shape
    x = dev(xyz)
    temp = torch.full((b, n), 1e10, dtype=torch.float32, device="cuda")
    idx = torch.zeros((b, m), dtype=torch.int32, device="cuda")
    if weights is None:
        pn.farthest_point_sampling_wrapper(b, n, m, x, temp, idx)
    else:
        pn.furthest_point_sampling_weights_wrapper(b, n, m, x, dev(weights), temp, idx)
    return idx.cpu().numpy()


@pytest.mark.parametrize("b,n,m", [(1, 16384, 4096), (2, 4096, 512), (3, 512, 256), (2, 1000, 100),
                                   (2, 100, 37), (1, 7, 7), (1, 1, 1), (2, 3000, 64), (1, 20000, 50)])
def test_fps_bit_exact(ext, oracle_ops, b, n, m):
    pn = ext[0]
    xyz = make_batch(10, b, n, dup_frac=0.1)[..., :3]
    np.testing.assert_array_equal(hip_fps(pn, xyz, m), oracle_ops.fps(xyz, m))


def test_fps_all_duplicates(ext, oracle_ops):
    """every point identical: the pick is decided purely by the tie rule (bitrev(k mod S), k)"""
    pn = ext[0]
    xyz = np.ones((1, 4096, 3), np.float32)
    np.testing.assert_array_equal(hip_fps(pn, xyz, 64), oracle_ops.fps(xyz, 64))
    xyz = np.ones((1, 300, 3), np.float32)
    np.testing.assert_array_equal(hip_fps(pn, xyz, 20), oracle_ops.fps(xyz, 20))


@pytest.mark.parametrize("b,n,m", [(2, 4096, 512), (2, 512, 256), (1, 16384, 300), (2, 777, 99), (1, 40, 9)])
def test_fps_weights_bit_exact(ext, oracle_ops, b, n, m):
    pn = ext[0]
    xyz = make_batch(20, b, n, dup_frac=0.1)[..., :3]
    rng = np.random.default_rng(5)
    w = (1.0 / (1.0 + np.exp(-rng.normal(size=(b, n)) * 3))).astype(np.float32)
    w[:, ::17] = 0.0          # exercises max(w, 1e-12) in double
    w[:, 5::29] = 1e-13
    w[:, 3::31] = w[:, 2:-1:31][:, :w[:, 3::31].shape[1]]  # equal weights -> ties
    np.testing.assert_array_equal(hip_fps(pn, xyz, m, w), oracle_ops.fps_weights(xyz, w, m))


def test_gather_and_group(ext, oracle_ops):
    pn = ext[0]
    rng = np.random.default_rng(1)
    b, c, n, m, ns = 2, 5, 300, 40, 16
    pts = rng.normal(size=(b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, (b, m)).astype(np.int32)
    out = torch.empty((b, c, m), device="cuda")
    pn.gather_points_wrapper(b, c, n, m, dev(pts), dev(idx), out)
    np.testing.assert_array_equal(out.cpu().numpy(), oracle_ops.gather_points(pts, idx))
    gidx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    gout = torch.empty((b, c, m, ns), device="cuda")
    pn.group_points_wrapper(b, c, n, m, ns, dev(pts), dev(gidx), gout)
    np.testing.assert_array_equal(gout.cpu().numpy(), oracle_ops.group_points(pts, gidx))
    # backward scatters: atomic order differs from the sequential oracle -> tolerance
    go = rng.normal(size=(b, c, m)).astype(np.float32)
    gp = torch.zeros((b, c, n), device="cuda")
    pn.gather_points_grad_wrapper(b, c, n, m, dev(go), dev(idx), gp)
    np.testing.assert_allclose(gp.cpu().numpy(), oracle_ops.gather_points_grad(go, idx, n), rtol=1e-5, atol=1e-5)
    ggo = rng.normal(size=(b, c, m, ns)).astype(np.float32)
    ggp = torch.zeros((b, c, n), device="cuda")
    pn.group_points_grad_wrapper(b, c, n, m, ns, dev(ggo), dev(gidx), ggp)
    np.testing.assert_allclose(ggp.cpu().numpy(), oracle_ops.group_points_grad(ggo, gidx, n), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,m,r_in,r_out,ns", [(16384, 512, 0.0, 0.2, 16), (16384, 512, 0.2, 0.8, 32),
                                               (4096, 1024, 0.8, 1.6, 32), (1024, 512, 1.6, 4.8, 32),
                                               (512, 256, 0.0, 6.4, 32), (100, 30, 0.0, 0.01, 8),
                                               (70, 9, 0.0, 100.0, 128)])
def test_ball_query_variants(ext, oracle_ops, n, m, r_in, r_out, ns):
    pn = ext[0]
    b = 2
    xyz = make_batch(30, b, n, dup_frac=0.05)[..., :3]
    new_xyz = np.ascontiguousarray(xyz[:, :m] + np.float32(0.01))
    new_xyz[:, 0] = 1000.0  # a centre with no neighbours at all
    x, q = dev(xyz), dev(new_xyz)
    # dilated
    cnt = torch.zeros((b, m), dtype=torch.int32, device="cuda")
    idx = torch.zeros((b, m, ns), dtype=torch.int32, device="cuda")
    pn.ball_query_dilated_wrapper(b, n, m, r_in, r_out, ns, q, x, cnt, idx)
    ocnt, oidx = oracle_ops.ball_query_dilated(r_in, r_out, ns, xyz, new_xyz)
    np.testing.assert_array_equal(cnt.cpu().numpy(), ocnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), oidx)
    # cnt
    cnt.zero_(); idx.zero_()
    pn.ball_query_cnt_wrapper(b, n, m, r_out, ns, q, x, cnt, idx)
    ocnt, oidx = oracle_ops.ball_query_cnt(r_out, ns, xyz, new_xyz)
    np.testing.assert_array_equal(cnt.cpu().numpy(), ocnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), oidx)
    # plain
    idx.zero_()
    pn.ball_query_wrapper(b, n, m, r_out, ns, q, x, idx)
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle_ops.ball_query(r_out, ns, xyz, new_xyz))


def test_three_nn_interpolate(ext, oracle_ops):
    pn = ext[0]
    b, n, m, c = 2, 1500, 700, 6
    unknown = make_batch(40, b, n)[..., :3]
    known = np.ascontiguousarray(unknown[:, :m])   # exact zeros + duplicates -> ties
    d2 = torch.empty((b, n, 3), device="cuda")
    idx = torch.empty((b, n, 3), dtype=torch.int32, device="cuda")
    pn.three_nn_wrapper(b, n, m, dev(unknown), dev(known), d2, idx)
    od2, oidx = oracle_ops.three_nn(unknown, known)
    np.testing.assert_array_equal(idx.cpu().numpy(), oidx)
    np.testing.assert_array_equal(d2.cpu().numpy(), od2)
    rng = np.random.default_rng(3)
    feats = rng.normal(size=(b, c, m)).astype(np.float32)
    w = rng.uniform(0, 1, (b, n, 3)).astype(np.float32)
    out = torch.empty((b, c, n), device="cuda")
    pn.three_interpolate_wrapper(b, c, m, n, dev(feats), idx, dev(w), out)
    np.testing.assert_array_equal(out.cpu().numpy(), oracle_ops.three_interpolate(feats, oidx, w))
    go = rng.normal(size=(b, c, n)).astype(np.float32)
    gp = torch.zeros((b, c, m), device="cuda")
    pn.three_interpolate_grad_wrapper(b, c, n, m, dev(go), idx, dev(w), gp)
    np.testing.assert_allclose(gp.cpu().numpy(), oracle_ops.three_interpolate_grad(go, oidx, w, m), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("k", [1, 2, 63, 64, 65, 256, 512, 700])
def test_iou_and_nms_bit_exact(ext, oracle_ops, k):
    _, nm, fused = ext
    boxes = random_boxes(k, k, spread=8.0 + k / 8)
    if k > 4:
        boxes[3] = boxes[1]                 # identical boxes
        boxes[4, 3:5] = 0.0                 # zero-area box
    bd = dev(boxes)
    iou = torch.zeros((k, k), device="cuda")
    nm.boxes_iou_bev_gpu(bd, bd, iou)
    np.testing.assert_array_equal(iou.cpu().numpy(), oracle_ops.boxes_iou_bev(boxes, boxes))
    ov = torch.zeros((k, k), device="cuda")
    nm.boxes_overlap_bev_gpu(bd, bd, ov)
    np.testing.assert_array_equal(ov.cpu().numpy(), oracle_ops.boxes_overlap_bev(boxes, boxes))
    for thr in (0.01, 0.1, 0.7):
        keep = torch.zeros(k, dtype=torch.int64)
        num = nm.nms_gpu(bd, keep, thr)
        np.testing.assert_array_equal(keep[:num].numpy(), oracle_ops.nms(boxes, thr))
        num = nm.nms_normal_gpu(bd, keep, thr)
        np.testing.assert_array_equal(keep[:num].numpy(), oracle_ops.nms(boxes, thr, normal=True))
    kd, nd = fused.nms_device(bd, 0.1)
    np.testing.assert_array_equal(kd[:int(nd.item())].cpu().numpy(), oracle_ops.nms(boxes, 0.1))


@pytest.mark.parametrize("rows,k,n", [(256, 16, 16), (1000, 132, 64), (384, 260, 256), (128, 68, 32),
                                      (4096, 512, 1024), (77, 4, 3), (640, 36, 96)])
def test_linear_rows_bit_exact(ext, oracle_ops, rows, k, n):
    fused = ext[2]
    rng = np.random.default_rng(rows + k)
    lda = (k + 3) // 4 * 4
    ldw = (n + 3) // 4 * 4
    a = rng.normal(size=(rows, lda)).astype(np.float32)
    w = (rng.normal(size=(k, ldw)) / np.sqrt(k)).astype(np.float32)
    shift = rng.normal(size=(n,)).astype(np.float32)
    for act in (0, 1):
        out = torch.zeros((rows, n + 5), device="cuda")
        fused.linear(dev(a), dev(w), dev(shift), act, out, k=k, ncols=n, col0=2)
        ref = np.zeros((rows, n + 5), np.float32)
        oracle_ops.linear(a, w[:, :n], shift, act, k=k, out=ref, col0=2)
        np.testing.assert_array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("n,m,ns,c,n1", [(512, 64, 16, 1, 16), (1024, 96, 32, 64, 64), (300, 40, 32, 128, 128),
                                         (256, 24, 16, 256, 256), (200, 10, 8, 5, 40)])
def test_linear_grouped_pool_bit_exact(ext, oracle_ops, n, m, ns, c, n1):
    fused = ext[2]
    b = 2
    rng = np.random.default_rng(n + m)
    ld = (3 + c + 3) // 4 * 4
    rows = np.zeros((b, n, ld), np.float32)
    rows[..., :3 + c] = rng.normal(size=(b, n, 3 + c))
    ctr = np.zeros((b, m, 4), np.float32)
    ctr[..., :3] = rng.normal(size=(b, m, 3))
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(0, 3, (b, m)).astype(np.int32)
    w = (rng.normal(size=(ld, n1)) / np.sqrt(ld)).astype(np.float32)
    w[3 + c:] = 0
    shift = rng.normal(size=(n1,)).astype(np.float32)
    # layer 1 unpooled (grouped gather + centre subtraction)
    out = torch.zeros((b * m * ns, n1), device="cuda")
    fused.linear(dev(rows), dev(w), dev(shift), 1, out, idx=dev(idx), ctr=dev(ctr))
    ref = oracle_ops.linear(rows, w, shift, 1, idx=idx, ctr=ctr)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # pooled + masked
    outp = torch.zeros((b * m, n1 + 4), device="cuda")
    fused.linear(dev(rows), dev(w), dev(shift), 1, outp, idx=dev(idx), ctr=dev(ctr), cnt=dev(cnt), pool=ns, col0=4)
    refp = np.zeros((b * m, n1 + 4), np.float32)
    oracle_ops.linear(rows, w, shift, 1, idx=idx, ctr=ctr, cnt=cnt, pool=ns, out=refp, col0=4)
    np.testing.assert_array_equal(outp.cpu().numpy(), refp)


def test_linear_random_shapes_bit_exact(ext, oracle_ops):
    """seeded sweep over shapes that land on every tile variant, interior and edge tiles, full / short / odd
    K slabs (the buffer-load fast path, its predicated first / last slab, both epilogues)"""
    fused = ext[2]
    rng = np.random.default_rng(2025)
    shapes = [(32768, 132, 128), (33000, 260, 256), (40000, 96, 128), (36864, 64, 96), (32896, 17, 40), (2048, 512, 128),
              (65600, 31, 300), (16384, 16, 1), (70000, 48, 33)]
    shapes += [(int(rng.integers(1, 40000)), int(rng.integers(1, 300)), int(rng.integers(1, 300))) for _ in range(10)]
    for rows, k, n in shapes:
        lda, ldw = (k + 3) // 4 * 4 + 4 * int(rng.integers(0, 2)), (n + 3) // 4 * 4
        a = rng.normal(size=(rows, lda)).astype(np.float32)
        a[:, k:] = np.nan                       # columns beyond K must never be read into the result
        w = (rng.normal(size=(k, ldw)) / np.sqrt(k)).astype(np.float32)
        shift = rng.normal(size=(n,)).astype(np.float32)
        act = int(rng.integers(0, 2))
        out = torch.full((rows, n + 3), 7.0, device="cuda")
        fused.linear(dev(a), dev(w), dev(shift), act, out, k=k, ncols=n, col0=1)
        ref = np.full((rows, n + 3), 7.0, np.float32)
        oracle_ops.linear(np.nan_to_num(a), w[:, :n], shift, act, k=k, out=ref, col0=1)
        np.testing.assert_array_equal(out.cpu().numpy(), ref, err_msg=str((rows, k, n, lda, act)))


def test_head_elementwise_bit_exact(ext, oracle_ops):
    fused = ext[2]
    rng = np.random.default_rng(9)
    r = 777
    s = (rng.normal(size=(r,)) * 4).astype(np.float32)
    for gamma in (1.0, 0.5, 2.0):
        np.testing.assert_array_equal(fused.sigmoid_pow(dev(s), gamma).cpu().numpy(), oracle_ops.sigmoid_pow(s, gamma))
    off = (rng.normal(size=(r, 4)) * 3).astype(np.float32)
    cand = rng.normal(size=(r, 4)).astype(np.float32) * 10
    vote = torch.zeros((r, 3), device="cuda")
    offo = torch.zeros((r, 3), device="cuda")
    fused.vote_points(dev(off), dev(cand), (3.0, 3.0, 2.0), vote, offo)
    ov, oo = oracle_ops.vote_points(off, cand, (3.0, 3.0, 2.0))
    np.testing.assert_array_equal(vote.cpu().numpy(), ov)
    np.testing.assert_array_equal(offo.cpu().numpy(), oo)
    code = rng.normal(size=(r, 32)).astype(np.float32)
    pts = rng.normal(size=(r, 3)).astype(np.float32) * 20
    thr, fac = np.float32(np.deg2rad(10.0)), np.float32(np.deg2rad(45.0))
    got = fused.decode_boxes(dev(code), dev(pts), 12, True, False, thr, fac).cpu().numpy()
    np.testing.assert_array_equal(got, oracle_ops.decode_boxes(code, pts))


@pytest.mark.parametrize("p,ncls", [(256, 1), (256, 3), (512, 1), (100, 2)])
def test_postprocess_bit_exact(ext, oracle_ops, p, ncls):
    fused = ext[2]
    b = 3
    rng = np.random.default_rng(p + ncls)
    cls = (rng.normal(size=(b * p, ncls)) * 2 - 1).astype(np.float32)
    cls[5] = cls[9]  # equal scores -> stable order
    boxes = np.zeros((b * p, 9), np.float32)
    boxes[:, :7] = random_boxes(1, b * p, spread=40.0)
    boxes[:, 7] = rng.normal(size=b * p) * 0.1
    cls[p:2 * p] = -10.0  # a scene with nothing above threshold
    got = fused.postprocess(dev(cls), dev(boxes), b, 0.1, 512, 100, 0.01)
    ref = oracle_ops.postprocess(cls, boxes, b, 0.1, 512, 100, 0.01)
    for g, r in zip(got, ref):
        np.testing.assert_array_equal(g.cpu().numpy(), r)
    assert ref[4][1] == 0 and ref[4][0] > 0


@pytest.mark.parametrize("p,kind", [(1024, "clusters"), (1024, "rows"), (1000, "dense"), (640, "clusters"), (8, "rows"), (1, "dense")])
def test_postprocess_pair_filtered_mask_at_1024_candidates(ext, oracle_ops, p, kind):
    """the two-phase suppression-mask kernel (separated pairs skipped, the rest evaluated from a dense queue: csrc/iou3d_nms.hip)
    at the candidate counts of BASELINE config 5 (1024 per scene) against the oracle, which evaluates EVERY pair: vote-like
    clusters (many near pairs per row), rows of collinear equal boxes (the rule's worst case), one dense blob (nothing skipped:
    the queue holds every pair of a chunk), no score filter so that all p candidates reach the NMS"""
    fused = ext[2]
    b = 3
    rng = np.random.default_rng(p * 7 + len(kind))
    boxes = np.zeros((b * p, 9), np.float32)
    boxes[:, :7] = random_boxes(3, b * p, spread=60.0)
    if kind == "clusters":
        centres = rng.uniform(-50, 50, (40, 2))
        pick = rng.integers(0, 40, b * p)
        boxes[:, 0] = centres[pick, 0] + rng.normal(size=b * p) * 0.7
        boxes[:, 1] = centres[pick, 1] + rng.normal(size=b * p) * 0.7
    elif kind == "rows":
        boxes[:, 3], boxes[:, 4], boxes[:, 6] = 3.9, 1.6, 0.0
        boxes[:, 0] = (np.arange(b * p) % 97) * 1.3
        boxes[:, 1] = ((np.arange(b * p) // 97) % 5) * 1.55
    else:
        boxes[:, 0] = rng.normal(size=b * p) * 2.0
        boxes[:, 1] = rng.normal(size=b * p) * 2.0
    boxes[:, 7] = rng.normal(size=b * p) * 0.1
    cls = (rng.normal(size=(b * p, 1)) * 2 + 1).astype(np.float32)
    for thr in (0.01, 0.3):
        got = fused.postprocess(dev(cls), dev(boxes), b, 0.0, 4096, 100, thr)
        ref = oracle_ops.postprocess(cls, boxes, b, 0.0, 4096, 100, thr)
        for g, r in zip(got, ref):
            np.testing.assert_array_equal(g.cpu().numpy(), r)
    assert ref[4].min() > 0


@pytest.mark.parametrize("n,m,sa,sb", [(16384, 700, (0.0, 0.2, 16), (0.2, 0.8, 32)), (4096, 1024, (0.0, 0.8, 16), (0.8, 1.6, 32)),
                                       (512, 256, (0.0, 4.8, 16), (0.0, 6.4, 32)), (100, 7, (0.0, 0.01, 8), (0.0, 50.0, 128)),
                                       (1000, 33, (1.0, 3.0, 5), (0.5, 2.0, 70))])
def test_ball_query_pair_matches_two_queries(ext, oracle_ops, n, m, sa, sb):
    fused = ext[2]
    b = 2
    xyz = make_batch(50, b, n, dup_frac=0.05)[..., :3]
    new_xyz = np.ascontiguousarray(xyz[:, :m] + np.float32(0.01))
    new_xyz[:, 0] = 1000.0
    oca, oia = oracle_ops.ball_query_dilated(sa[0], sa[1], sa[2], xyz, new_xyz)
    ocb, oib = oracle_ops.ball_query_dilated(sb[0], sb[1], sb[2], xyz, new_xyz)
    for grid in (False, True):   # brute-force sweep and grid-hashed search give the same lists
        ca, ia, cb, ib = fused.ball_query_pair(dev(xyz), dev(new_xyz), sa, sb, grid=grid)
        np.testing.assert_array_equal(ca.cpu().numpy(), oca)
        np.testing.assert_array_equal(ia.cpu().numpy(), oia)
        np.testing.assert_array_equal(cb.cpu().numpy(), ocb)
        np.testing.assert_array_equal(ib.cpu().numpy(), oib)


@pytest.mark.parametrize("n,m", [(4096, 512), (512, 256), (16384, 700), (2048, 300), (16384, 2048), (4096, 4096)])
def test_weighted_sampler_fp32_scoring_and_its_exact_double_fallback(ext, oracle_ops, n, m):
    """S-FPS through det6d_fps_fused (round 5): scenes whose weights are all >= 1e-12 are scored with one fp32 multiply per
    point (exactly float(double(t) * double(w)): the product of two floats is exact in double); a scene holding a weight below
    1e-12 (sigmoid(score) ** gamma underflowing it) or a NaN weight hands itself over to the exact-double launch behind.
    Scenes of both kinds in ONE call, against the oracle's restatement of sampling_gpu.cu:419-540.  From round 6 the 16384- and
    4096-point clouds go through the MULTI-PICK score-weighted kernel (fps_seq.hip: fps_seq_w_kernel; (16384, 2048) is the second
    layer of the 65536-point configuration, (4096, 4096) samples every point): records by score, the region bound on scores, the
    box test on min-distances, pick 0 = arg-max of the weights in the reference's tie order, scene 4 = ties everywhere."""
    fused = ext[2]
    b = 5
    pts = make_batch(170 + n, b, n, dup_frac=0.1)
    xyz = np.ascontiguousarray(pts[..., :3])
    rng = np.random.default_rng(n + m)
    scores = (rng.normal(size=(b, n)) * 3).astype(np.float32)
    scores[1, rng.choice(n, 40, replace=False)] = -40.0          # sigmoid -> 4e-18 < 1e-12: clamped in double by the reference
    scores[2, 17] = np.nan                                        # max(NaN, 1e-12) = 1e-12 (sampling_gpu.cu:466)
    scores[3] = -60.0                                             # every weight below the clamp
    scores[4, :] = 50.0                                           # every weight 1.0: ties everywhere -> the reference's tie order
    for gamma in (1.0, 2.5):
        got = torch.full((b, m + 3), -5, dtype=torch.int32, device="cuda")
        fused.fps_fused(dev(xyz), 0, n, m, dev(scores), gamma, got, 3)
        want = np.full((b, m + 3), -5, np.int32)
        oracle_ops.fps_fused(xyz, 0, n, m, scores, gamma, want, 3)
        np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_fused_sampler_and_helpers(ext, oracle_ops):
    """det6d_fps_fused (range slice + sigmoid**gamma + 1e10 init + offset), gather_centres, with_batch_index, pack"""
    fused = ext[2]
    b, n = 2, 4096
    pts = make_batch(70, b, n, dup_frac=0.1)
    xyz = np.ascontiguousarray(pts[..., :3])
    rng = np.random.default_rng(3)
    scores = (rng.normal(size=(b, n)) * 3).astype(np.float32)
    for lo, hi, m, sc, gamma, off in [(0, 4096, 512, None, 1.0, 0), (0, 4096, 512, scores, 1.0, 512), (512, 1024, 256, None, 1.0, 0),
                                      (0, 512, 256, scores, 0.5, 256), (100, 3100, 77, scores, 2.0, 3)]:
        got = torch.full((b, 1100), -5, dtype=torch.int32, device="cuda")
        fused.fps_fused(dev(xyz), lo, hi, m, dev(sc) if sc is not None else None, gamma, got, off)
        want = np.full((b, 1100), -5, np.int32)
        oracle_ops.fps_fused(xyz, lo, hi, m, sc, gamma, want, off)
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        # and it equals the unfused reference sequence
        sl = np.ascontiguousarray(xyz[:, lo:hi])
        ref = oracle_ops.fps(sl, m) if sc is None else oracle_ops.fps_weights(sl, oracle_ops.sigmoid_pow(sc[:, lo:hi], gamma), m)
        np.testing.assert_array_equal(want[:, off:off + m], ref + lo)
    idx = rng.integers(0, n, (b, 300)).astype(np.int32)
    rows = torch.full((b, 300, 8), 9.0, device="cuda")
    cx = fused.gather_centres(dev(xyz), dev(idx), rows, 6)
    ox, orows = oracle_ops.gather_centres(xyz, idx, 8, 6)
    np.testing.assert_array_equal(cx.cpu().numpy(), ox)
    r = rows.cpu().numpy()
    np.testing.assert_array_equal(r[..., :3], ox)
    assert (r[..., 3:6] == 9.0).all() and (r[..., 6:] == 0.0).all()
    np.testing.assert_array_equal(fused.with_batch_index(cx, 3).cpu().numpy(), oracle_ops.with_batch_index(ox, 3))
    flat = np.concatenate([np.repeat(np.arange(b, dtype=np.float32), n)[:, None], pts.reshape(b * n, 4)], 1)
    rows2, xyz2 = fused.pack_points(dev(flat), 4)
    orows2, oxyz2 = oracle_ops.pack_points(flat, 4)
    np.testing.assert_array_equal(rows2.cpu().numpy(), orows2)
    np.testing.assert_array_equal(xyz2.cpu().numpy(), oxyz2)


def test_pruned_fps_kernel_is_exact():
    """fps_seq.hip on fps_cells.hip's k-d regions: the multi-pick sampler (16384 points: 16 waves x 16 slots, several picks
    per barrier round) gives the oracle's picks bit for bit, duplicates and all-equal clouds included; the other sizes of the
    first script run the fat-thread kernels.  The one-pick wave-skip sampler of rounds 2-3 (experiments build) likewise."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_scripts", "fps_cells.py")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if "exact" in l]
    assert len(lines) == 7 and all("exact=True" in l or "exact True" in l for l in lines), out.stdout
    # the 16384-point suites of the round-4 sampler work: ray-cast scenes, every point twice, a lattice of exact ties, far
    # outliers, 32 scenes, m = n
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_scripts", "fps_seq.py")],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ALL EXACT" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    # ... and with the LDS of every CU filled with 0x7F7F0000 (a huge float, an index with high bits set) right before the sampling kernel (experiments build: same kernel
    # source plus the fill hook, DET6D_DBG_POISON_LDS): a sampler must not read LDS it has not written
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_scripts", "fps_seq.py")],
                         capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, DET6D_EXPERIMENTS_LIB="1", DET6D_DBG_POISON_LDS="0x7F7F0000"))
    assert out.returncode == 0 and "ALL EXACT" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_scripts", "fps_seq.py")],
                         capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, DET6D_EXPERIMENTS_LIB="1", DET6D_FPS_SEQ="0", DET6D_DBG_POISON_LDS="0x7F7F0000"))
    assert out.returncode == 0 and "ALL EXACT" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_skip_sampler_on_adversarial_clouds():
    """16384-point sampler (fps_seq.hip): more seeds, a lattice with thousands of exact distance ties, collinear points, far
    outliers — always the oracle's indices"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_scripts", "fps_stress.py")], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ALL True" in out.stdout, out.stdout


@pytest.mark.parametrize("lda,c1", [(68, 64), (132, 128), (260, 256), (8, 32)])
def test_group_expand_equals_the_gathered_first_layer(ext, oracle_ops, lda, c1):
    """csrc/expand.hip: per-point partial sums (one plain GEMM over the points, coordinate rows zeroed) + three FMAs per
    grouped output == the oracle's first layer over the gathered rows (feature columns first, relative coordinates
    last), dense rows and compact lists, bit for bit"""
    fused = ext[2]
    rng = np.random.default_rng(lda)
    b, n, m, ns = 2, 700, 96, 16
    rows = rng.normal(size=(b, n, lda)).astype(np.float32)
    rows[..., lda - 1] = 0.0                                    # pad column
    ctr = np.ascontiguousarray(rows[:, :m, :3] + rng.normal(size=(b, m, 3)).astype(np.float32) * 0.1)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    w = (rng.normal(size=(lda, c1)) * 0.2).astype(np.float32)
    w[lda - 1] = 0.0
    shift = rng.normal(size=(c1,)).astype(np.float32)
    ref = oracle_ops.linear(rows, w, shift, 1, idx=idx, ctr=ctr)
    wz = w.copy(); wz[:3] = 0.0
    p = torch.empty((b * n, c1), device='cuda')
    fused.linear(dev(rows).view(b * n, lda), dev(wz), None, 0, p)
    out = torch.full((b * m * ns, c1 + 4), 7.0, device='cuda')
    fused.group_expand(p, 0, dev(w), dev(shift), 1, c1, dev(rows), dev(ctr), out, idx=dev(idx))
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got[:, :c1], ref)
    assert (got[:, c1:] == 0).all()
    # the gathered GEMM of det6d_linear follows the same chain order
    y = torch.empty((b * m * ns, c1), device='cuda')
    fused.linear(dev(rows), dev(w), dev(shift), 1, y, idx=dev(idx), ctr=dev(ctr))
    np.testing.assert_array_equal(y.cpu().numpy(), ref)
    # compact list of the same neighbourhoods: every live row equals the dense row it stands for
    cnt = rng.integers(0, ns + 1, (b, m)).astype(np.int32)
    for bi in range(b):
        for j in range(m):
            c = max(int(cnt[bi, j]), 1)
            idx[bi, j] = np.resize(idx[bi, j, :c], ns)            # cyclic padding like the ball query's
    cr = fused.compact_groups(dev(cnt), dev(idx), n)
    outc = torch.full((cr.capacity, c1), 5.0, device='cuda')
    fused.group_expand(p, 0, dev(w), dev(shift), 1, c1, dev(rows), dev(ctr), outc, compact=cr)
    yc = torch.empty((cr.capacity, c1), device='cuda')
    fused.linear(dev(rows), dev(w), dev(shift), 1, yc, ctr=dev(ctr), compact=cr, gather=True)
    live = int(cr.hdr[0].item())
    tags = cr.crow_c[:live].cpu().numpy()
    a, bb = outc[:live].cpu().numpy(), yc[:live].cpu().numpy()
    np.testing.assert_array_equal(a[tags >= 0], bb[tags >= 0])
    assert (a[tags < 0] == 0).all()


def test_cooperative_sampler_for_large_scenes():
    """csrc/fps_coop.hip: D-FPS of 32768 / 65536-point scenes held in registers by 2 / 4 cooperating workgroups per scene
    (BASELINE config 5): the oracle's picks bit for bit, ties / duplicates / odd batch sizes included; and the
    memory-resident fallback (a scratch too small for the cooperative workspace) still agrees; so does the opt-in same-XCD fast path"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_scripts", "fps_coop.py")], capture_output=True, text=True,
                         timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ALL True" in out.stdout, out.stdout
    print(out.stdout)
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_scripts", "fps_coop.py"), "fallback"], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ALL True" in out.stdout and "fallback" in out.stdout, out.stdout
    # DET6D_FPS_COOP_FAST=1: workgroup-scope publishing stores where the placement test and the round-0 handshake allow (the
    # default publishes with agent-scope stores: the memory model's guarantee)
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_scripts", "fps_coop.py")], capture_output=True, text=True,
                         timeout=1500, env=dict(os.environ, DET6D_FPS_COOP_FAST="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ALL True" in out.stdout and "fallback" not in out.stdout, out.stdout
    print(out.stdout)
    # the LDS of every CU filled with 0x7F7F0000 (a huge float, an index with high bits set) right before the sampling kernel (experiments build: same kernel source plus
    # the fill hook, DET6D_DBG_POISON_LDS).  Round 4: a record slot never written in a launch leaked into the published candidate count — invisible in a
    # fresh process, where the LDS is zero, and wrong picks in a long-lived one.
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_scripts", "fps_coop.py")], capture_output=True,
                         text=True, timeout=1500, env=dict(os.environ, DET6D_EXPERIMENTS_LIB="1", DET6D_DBG_POISON_LDS="0x7F7F0000"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ALL True" in out.stdout, out.stdout


def test_cooperative_sampler_call_over_more_scenes_than_the_chip_holds(ext, oracle_ops):
    """one det6d_fps_fused call over 72 scenes of 65536 points = 288 cooperating workgroups on a 256-CU chip: the call samples
    them in chunks of one workgroup per CU (round-4 review: a C-ABI caller could starve itself until the 2 s time-out); picks
    equal the oracle's on a few scenes of every chunk, the status word stays clear"""
    fused = ext[2]
    b, n, m = 72, 65536, 48
    rng = np.random.default_rng(99)
    base = make_batch(640, 3, n)[..., :3]
    xyz = np.ascontiguousarray(base[np.arange(b) % 3] + rng.normal(size=(b, 1, 3)).astype(np.float32) * np.float32(0.01))
    idx = torch.full((b, m), -1, dtype=torch.int32, device="cuda")
    ws = fused.fps_workspace(b, n)
    fused.fps_fused(dev(xyz), 0, n, m, None, 1.0, idx, 0, temp=ws)
    fused.fps_status(b, n, ws)                                   # raises if a workgroup gave up waiting for its partners
    got = idx.cpu().numpy()
    assert (got >= 0).all() and (got < n).all() and (got[:, 0] == 0).all()
    for s in (0, 1, 63, 64, 65, 71):
        np.testing.assert_array_equal(got[s], oracle_ops.fps(xyz[s:s + 1], m)[0], err_msg="scene %d" % s)


def test_ball_query_grid_adversarial(ext, oracle_ops):
    """grid search corner cases: centres outside the cloud, everything in one cell, outliers that stretch
    the bounding box past 128 cells, vertical stacks (the grid is 2-D), non-finite points"""
    fused = ext[2]
    rng = np.random.default_rng(11)
    n, m = 3000, 200
    base = (rng.normal(size=(1, n, 3)) * [8, 8, 1]).astype(np.float32)
    clouds = []
    c0 = base.copy(); clouds.append(c0)
    c1 = base.copy(); c1[0, :5, 0] = [500.0, -400.0, 300.0, 250.0, -350.0]; clouds.append(c1)      # far outliers
    c2 = (base * [0.01, 0.01, 5.0]).astype(np.float32); clouds.append(c2)                         # one cell, tall
    c3 = base.copy(); c3[0, 7] = np.nan; c3[0, 9, 1] = np.inf; clouds.append(c3)
    for xyz in clouds:
        new_xyz = np.ascontiguousarray(xyz[:, :m] + np.float32(0.05))
        new_xyz[0, 0] = (100.0, 100.0, 0.0)
        new_xyz[0, 1] = xyz[0, :, :].min(0) - 0.3 if np.isfinite(xyz).all() else (0, 0, 0)
        for sa, sb in [((0.0, 0.8, 16), (0.8, 1.6, 32)), ((0.0, 4.8, 16), (0.0, 6.4, 32)), ((0.0, 0.05, 8), (0.05, 0.3, 8))]:
            oca, oia = oracle_ops.ball_query_dilated(sa[0], sa[1], sa[2], xyz, new_xyz)
            ocb, oib = oracle_ops.ball_query_dilated(sb[0], sb[1], sb[2], xyz, new_xyz)
            ca, ia, cb, ib = fused.ball_query_pair(dev(xyz), dev(new_xyz), sa, sb, grid=True)
            np.testing.assert_array_equal(ca.cpu().numpy(), oca)
            np.testing.assert_array_equal(ia.cpu().numpy(), oia)
            np.testing.assert_array_equal(cb.cpu().numpy(), ocb)
            np.testing.assert_array_equal(ib.cpu().numpy(), oib)


def test_ball_query_grid_dense_shells(ext, oracle_ops):
    """shells with hundreds of hits per centre (the dense part of a real sweep): the running selection of the nsample
    smallest hit indices is cut back several times per centre (csrc/ball_query_grid.hip: keep_smallest + threshold); point
    order ascending, descending and shuffled inside the cells"""
    fused = ext[2]
    from tests.util import beam_batch
    rng = np.random.default_rng(5)
    n, m = 16384, 900
    beam = beam_batch(4500, 1, n)[..., :3]
    blob = (rng.normal(size=(1, n, 3)) * [2.0, 2.0, 0.3]).astype(np.float32)            # ~1000 points per 0.8 m cell
    clouds = [beam, blob, np.ascontiguousarray(blob[:, ::-1]), np.ascontiguousarray(blob[:, np.argsort(blob[0, :, 0])])]
    for xyz in clouds:
        new_xyz = np.ascontiguousarray(xyz[:, rng.choice(n, m, replace=False)] + np.float32(0.003))
        for sa, sb in [((0.0, 0.2, 16), (0.2, 0.8, 32)), ((0.0, 0.8, 16), (0.0, 0.8, 64))]:
            oca, oia = oracle_ops.ball_query_dilated(sa[0], sa[1], sa[2], xyz, new_xyz)
            ocb, oib = oracle_ops.ball_query_dilated(sb[0], sb[1], sb[2], xyz, new_xyz)
            assert ocb.max() == sb[2]
            ca, ia, cb, ib = fused.ball_query_pair(dev(xyz), dev(new_xyz), sa, sb, grid=True)
            np.testing.assert_array_equal(ca.cpu().numpy(), oca)
            np.testing.assert_array_equal(ia.cpu().numpy(), oia)
            np.testing.assert_array_equal(cb.cpu().numpy(), ocb)
            np.testing.assert_array_equal(ib.cpu().numpy(), oib)


@pytest.mark.parametrize("c_in,widths,ns", [(1, (16, 16, 32), 16), (1, (32, 32, 64), 32), (4, (24, 32, 40), 16), (1, (8, 16, 16), 32),
                                             (64, (64, 64, 128), 16), (64, (64, 96, 128), 32), (64, (64, 96, 128), 16), (64, (64, 64, 128), 32)])
def test_mlp_chain3_equals_three_linears(ext, oracle_ops, c_in, widths, ns):
    """the fused narrow-MLP launch against the oracle's three-layer sequence (and hence against three
    det6d_linear calls, which test_linear_* pins to the same oracle)"""
    fused = ext[2]
    b, n, m = 2, 700, 96
    rng = np.random.default_rng(sum(widths))
    ld = (3 + c_in + 3) // 4 * 4
    rows = np.zeros((b, n, ld), np.float32)
    rows[..., :3 + c_in] = rng.normal(size=(b, n, 3 + c_in))
    ctr = rng.normal(size=(b, m, 3)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    cnt = rng.integers(0, 3, (b, m)).astype(np.int32)
    dims = [ld] + list(widths)
    layers_np, layers_dev = [], []
    for i in range(3):
        kin = dims[i] if i == 0 else (dims[i] + 3) // 4 * 4
        wpad = (dims[i + 1] + 3) // 4 * 4
        w = np.zeros((kin, wpad), np.float32)
        w[:dims[i] if i else 3 + c_in, :dims[i + 1]] = rng.normal(size=(dims[i] if i else 3 + c_in, dims[i + 1])) / np.sqrt(dims[i])
        s = rng.normal(size=(dims[i + 1],)).astype(np.float32)
        layers_np.append((w, s))
        layers_dev.append((dev(w), dev(s), dims[i + 1], 1))
    assert fused.chain_eligible(ld, layers_dev, ns)
    out = torch.zeros((b * m, widths[2] + 3), device="cuda")
    fused.mlp_chain3(dev(rows), dev(idx), dev(ctr), dev(cnt), layers_dev, out, 3)
    h = oracle_ops.linear(rows, layers_np[0][0], layers_np[0][1], 1, idx=idx, ctr=ctr)
    h = oracle_ops.linear(np.ascontiguousarray(np.pad(h, ((0, 0), (0, layers_np[1][0].shape[0] - h.shape[1])))), layers_np[1][0], layers_np[1][1], 1)
    h = np.ascontiguousarray(np.pad(h[:, :widths[1]], ((0, 0), (0, layers_np[2][0].shape[0] - widths[1]))))
    ref = np.zeros((b * m, widths[2] + 3), np.float32)
    oracle_ops.linear(h, layers_np[2][0][:, :widths[2]], layers_np[2][1], 1, cnt=cnt, pool=ns, out=ref, col0=3)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_fallback_kernels_via_env_switches(tmp_path):
    """the plain predicated GEMM loader (tensors beyond 32-bit offsets), the LDS chain kernel and the
    three-GEMM route for SA2-sized groups are selected by size / shape at run time; force them with their
    environment switches in a child process and rerun the parity tests that cover them"""
    import subprocess
    import sys
    env = dict(os.environ, DET6D_EXPERIMENTS_LIB='1', DET6D_LINEAR_NO_FAST='1', DET6D_CHAIN_LDS='1', DET6D_CHAIN_NO_WIDE='1')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_ops_gpu.py'), '-q', '-x', '-m', 'gpu',
                          '-k', 'test_linear or (chain3 and not widths4 and not widths5 and not widths6 and not widths7)'], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'passed' in out.stdout


@pytest.mark.parametrize("b,n,m,c", [(2, 16384, 4096, 64), (1, 300, 50, 3), (1, 2048, 16384, 5), (3, 1024, 4093, 7), (2, 5000, 512, 4)])
def test_three_interpolate_shapes(ext, oracle_ops, b, n, m, c):
    """both forms of det6d_three_interpolate (channel rows staged in LDS when four of them fit 64 KB and n >= 1024, the plain
    gather otherwise; odd m: unaligned rows) against the oracle, bit for bit"""
    pn = ext[0]
    rng = np.random.default_rng(n + m)
    feats = rng.normal(size=(b, c, m)).astype(np.float32)
    idx = rng.integers(0, m, size=(b, n, 3)).astype(np.int32)
    w = rng.uniform(0, 1, (b, n, 3)).astype(np.float32)
    out = torch.empty((b, c, n), device="cuda")
    pn.three_interpolate_wrapper(b, c, m, n, dev(feats), dev(idx), dev(w), out)
    np.testing.assert_array_equal(out.cpu().numpy(), oracle_ops.three_interpolate(feats, idx, w))
