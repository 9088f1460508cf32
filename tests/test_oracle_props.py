"""Properties and known answers of the CPU oracle's index ops (the reference has no CPU code or
tests for them, so these pin the restatement's own semantics: SURVEY.md A.2)."""
import numpy as np
import pytest

from tests.util import make_batch


def bitrev(v, bits):
    return int(format(v, '0%db' % bits)[::-1], 2) if bits else 0


def fps_by_rule(xyz, m, weights=None):
    """independent FPS: plain argmax + the closed-form tie rule 'minimise (bitrev(k mod S), k)'"""
    n = xyz.shape[0]
    S = min(1 << int(np.floor(np.log2(n))), 1024)
    bits = int(np.log2(S))
    key = np.array([bitrev(k % S, bits) * (1 << 32) + k for k in range(n)], np.int64)
    temp = np.full(n, 1e10, np.float32)
    out = []

    def pick(score):
        best = score.max()
        if not best > -1.0:
            return 0
        cand = np.where(score == best)[0]
        return int(cand[np.argmin(key[cand])])

    if weights is None:
        old = 0
        out.append(0)
        rounds = m - 1
    else:
        old = pick(weights.astype(np.float32))
        out.append(old)
        rounds = m - 1
    for _ in range(rounds):
        d = xyz.astype(np.float32) - xyz[old].astype(np.float32)
        dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
        t = (dy * dy).astype(np.float32)
        t = (np.float64(dx) * np.float64(dx) + np.float64(t)).astype(np.float32)      # fma(dx,dx,dy*dy)
        t = (np.float64(dz) * np.float64(dz) + np.float64(t)).astype(np.float32)      # fma(dz,dz,.)
        temp = np.minimum(temp, t)
        score = temp if weights is None else (temp.astype(np.float64) * np.maximum(weights.astype(np.float64), 1e-12)).astype(np.float32)
        old = pick(score)
        out.append(old)
    return np.array(out, np.int32)


@pytest.mark.parametrize("n,m", [(8, 8), (64, 20), (100, 30), (512, 64), (1024, 40), (3000, 50)])
def test_fps_literal_block_simulation_equals_tie_rule(oracle_ops, n, m):
    xyz = make_batch(3, 1, n, dup_frac=0.3)[0, :, :3]
    xyz[n // 2:] = xyz[:n - n // 2]          # every point has an exact duplicate
    np.testing.assert_array_equal(oracle_ops.fps(xyz[None], m)[0], fps_by_rule(xyz, m))


def test_fps_tie_example_from_survey(oracle_ops):
    # S = 8: a tie between k = 1 and k = 8 is won by k = 8 (bitrev(0) < bitrev(1))
    xyz = np.zeros((1, 9, 3), np.float32)
    xyz[0, 1] = xyz[0, 8] = (1, 0, 0)
    assert oracle_ops.opt_n_threads(9) == 8
    assert list(oracle_ops.fps(xyz, 2)[0]) == [0, 8]


@pytest.mark.parametrize("n,m", [(64, 16), (500, 60), (2048, 64)])
def test_fps_weights_rule(oracle_ops, n, m):
    rng = np.random.default_rng(n)
    xyz = make_batch(4, 1, n, dup_frac=0.2)[0, :, :3]
    w = rng.uniform(0, 1, n).astype(np.float32)
    w[::7] = 0.0
    w[3::11] = w[2:-1:11][:len(w[3::11])]
    np.testing.assert_array_equal(oracle_ops.fps_weights(xyz[None], w[None], m)[0], fps_by_rule(xyz, m, w))


def test_opt_n_threads(oracle_ops):
    for n, want in [(1, 1), (2, 2), (3, 2), (7, 4), (8, 8), (1000, 512), (1024, 1024), (16384, 1024), (4096, 1024), (512, 512)]:
        assert oracle_ops.opt_n_threads(n) == want


def test_ball_query_semantics(oracle_ops):
    xyz = np.zeros((1, 10, 3), np.float32)
    xyz[0, :, 0] = np.arange(10)
    q = np.array([[[4.0, 0, 0], [100.0, 0, 0]]], np.float32)
    cnt, idx = oracle_ops.ball_query_cnt(1.5, 5, xyz, q)
    assert list(cnt[0]) == [3, 0]
    assert list(idx[0, 0]) == [3, 4, 5, 3, 4]      # cyclic repetition of the hits
    assert list(idx[0, 1]) == [0, 0, 0, 0, 0]      # untouched zeros
    assert list(oracle_ops.ball_query(1.5, 5, xyz, q)[0, 0]) == [3, 4, 5, 3, 3]   # pad with the first hit
    cnt, idx = oracle_ops.ball_query_dilated(1.0, 2.5, 4, xyz, q)   # 1 <= d < 2.5  (d2 in [1, 6.25))
    assert list(cnt[0]) == [4, 0] and list(idx[0, 0]) == [2, 3, 5, 6]
    cnt, idx = oracle_ops.ball_query_cnt(100.0, 4, xyz, q)           # early stop at nsample
    assert list(idx[0, 0]) == [0, 1, 2, 3] and cnt[0, 0] == 4


def test_three_nn_and_interpolate(oracle_ops):
    known = np.array([[[0, 0, 0], [1, 0, 0], [0, 2, 0], [5, 5, 5], [0, 0, 0]]], np.float32)
    unknown = np.array([[[0, 0, 0], [0.9, 0, 0]]], np.float32)
    d2, idx = oracle_ops.three_nn(unknown, known)
    assert list(idx[0, 0]) == [0, 4, 1]            # strict '<': the earlier of two equal points wins
    np.testing.assert_allclose(d2[0, 0], [0, 0, 1])
    feats = np.arange(10, dtype=np.float32).reshape(1, 2, 5)
    w = np.array([[[0.5, 0.25, 0.25], [1, 0, 0]]], np.float32)
    out = oracle_ops.three_interpolate(feats, idx, w)
    np.testing.assert_allclose(out[0, 0, 0], 0.5 * 0 + 0.25 * 4 + 0.25 * 1)


def test_nms_properties(oracle_ops):
    from tests.util import random_boxes
    boxes = random_boxes(5, 200, spread=15.0)
    keep = oracle_ops.nms(boxes, 0.1)
    assert keep[0] == 0 and np.all(np.diff(keep) > 0)
    iou = oracle_ops.boxes_iou_bev(boxes, boxes)
    sub = iou[np.ix_(keep, keep)]
    assert (np.triu(sub, 1) <= 0.1).all()                       # survivors do not suppress each other
    np.testing.assert_array_equal(oracle_ops.nms(boxes[keep], 0.1), np.arange(len(keep)))  # idempotent
    np.testing.assert_allclose(np.diag(iou), 1.0, atol=1e-5)
    np.testing.assert_allclose(iou, iou.T, atol=1e-5)
    assert len(oracle_ops.nms(np.zeros((0, 7), np.float32), 0.1)) == 0


def test_linear_matches_numpy(oracle_ops):
    rng = np.random.default_rng(0)
    a = rng.normal(size=(50, 12)).astype(np.float32)
    w = rng.normal(size=(12, 7)).astype(np.float32)
    s = rng.normal(size=7).astype(np.float32)
    got = oracle_ops.linear(a, w, s, 1)
    np.testing.assert_allclose(got, np.maximum(a.astype(np.float64) @ w + s, 0), rtol=1e-5, atol=1e-5)
    idx = rng.integers(0, 20, (2, 3, 16)).astype(np.int32)
    rows = rng.normal(size=(2, 20, 12)).astype(np.float32)
    ctr = rng.normal(size=(2, 3, 3)).astype(np.float32)
    cnt = np.array([[1, 0, 2], [0, 3, 1]], np.int32)
    got = oracle_ops.linear(rows, w, s, 1, idx=idx, ctr=ctr, cnt=cnt, pool=16)
    g = rows[np.arange(2)[:, None, None], idx].astype(np.float64)
    g[..., :3] -= ctr[:, :, None, :]
    want = np.maximum(g @ w + s, 0).max(axis=2) * (cnt > 0)[..., None]
    np.testing.assert_allclose(got.reshape(2, 3, 7), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("smin,split", [(1, 1), (1, 4), (4, 4), (4, 0), (2, 2)])
def test_compact_rows_equal_dense_rows(oracle_ops, smin, split):
    """The claim behind de6d_amd/csrc/compact.hip, checked on the CPU alone: a ball query pads a ball of cnt < nsample
    hits with repetitions of its first cnt hits (ball_query_gpu.cu:75-90,114-129), so the grouped MLP + max-pool over
    the compact row list (first cnt slots, cut into power-of-two parts) equals the MLP over ALL nsample rows bit for
    bit — on the indices the oracle's own dilated ball query produces."""
    rng = np.random.default_rng(5 + smin + split)
    b, n, m, ns, c_in = 2, 600, 300, 8, 5
    xyz = rng.uniform(-2, 2, (b, n, 3)).astype(np.float32)
    ctr = xyz[:, rng.choice(n, m, replace=False)].copy()
    cnt, idx = oracle_ops.ball_query_dilated(0.25, 0.55, ns, xyz, ctr)
    assert 0 < (cnt == 0).sum() and (cnt >= ns).sum() > 0 and (cnt % 4 != 0).sum() > 0
    ld = (3 + c_in + 3) // 4 * 4
    rows = np.zeros((b, n, ld), np.float32)
    rows[..., :3] = xyz
    rows[..., 3:3 + c_in] = rng.normal(size=(b, n, c_in))
    widths, layers, kin = (12, 20, 24), [], ld
    for cout in widths:
        w = np.zeros((kin, cout), np.float32)
        w[:, :] = rng.normal(size=(kin, cout)) / np.sqrt(kin)
        layers.append((w, rng.normal(size=(cout,)).astype(np.float32)))
        kin = cout
    h = oracle_ops.linear(rows, layers[0][0], layers[0][1], 1, idx=idx, ctr=ctr)
    h = oracle_ops.linear(h, layers[1][0], layers[1][1], 1)
    dense = oracle_ops.linear(h, layers[2][0], layers[2][1], 1, cnt=cnt, pool=ns)
    lists = oracle_ops.compact_groups(cnt, idx, n, smin=smin, split=max(split, smin) if split else 0)
    assert lists[0][8] == np.minimum(cnt, ns).sum() and lists[0][0] < b * m * ns
    compact = oracle_ops.mlp_chain3_compact(rows, lists, ctr, layers, np.zeros((b * m, widths[2]), np.float32))
    np.testing.assert_array_equal(compact, dense)


@pytest.mark.parametrize("lda,c1", [(8, 32), (68, 64), (260, 256)])
def test_first_layer_from_per_point_sums_is_the_gathered_chain(oracle_ops, lda, c1):
    """The chain order of gathered rows (features first, relative coordinates last: chain_k) makes the feature part of a
    grouped MLP's first layer a function of the point alone.  On the CPU, bit for bit: one plain layer over the POINTS
    with the coordinate rows of W zeroed, then three FMAs per grouped output (det6d_oracle_group_expand = csrc/expand.hip
    restated) == det6d_oracle_linear over the gathered rows — dense rows and compact lists."""
    rng = np.random.default_rng(lda)
    b, n, m, ns = 2, 300, 40, 16
    rows = rng.normal(size=(b, n, lda)).astype(np.float32)
    rows[..., lda - 1] = 0.0
    ctr = np.ascontiguousarray(rows[:, :m, :3] + 0.1 * rng.normal(size=(b, m, 3)).astype(np.float32))
    w = (0.2 * rng.normal(size=(lda, c1))).astype(np.float32)
    shift = rng.normal(size=(c1,)).astype(np.float32)
    cnt = rng.integers(0, ns + 1, (b, m)).astype(np.int32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    for bi in range(b):
        for j in range(m):
            idx[bi, j] = np.resize(idx[bi, j, :max(int(cnt[bi, j]), 1)], ns)
    want = oracle_ops.linear(rows, w, shift, 1, idx=idx, ctr=ctr)                      # the gathered first layer
    wz = w.copy()
    wz[:3] = 0.0
    p = oracle_ops.linear(rows.reshape(b * n, lda), wz, None, 0)                       # per-point sums
    got = oracle_ops.group_expand(p, 0, w, shift, 1, c1, rows, ctr, c1 + 4, idx=idx)
    np.testing.assert_array_equal(got[:, :c1], want)
    assert (got[:, c1:] == 0).all()
    lists = oracle_ops.compact_groups(cnt, idx, n)
    gotc = oracle_ops.group_expand(p, 0, w, shift, 1, c1, rows, ctr, c1, lists=lists)
    hdr, crow_p, crow_c = lists
    live = int(hdr[0])
    for r in range(live):
        tag = int(crow_c[r])
        if tag < 0:
            assert (gotc[r] == 0).all()
            continue
        cj, prow = tag & 0x1fffffff, int(crow_p[r])
        # the dense row of the same (centre, point) pair
        slot = int(np.where(idx.reshape(b * m, ns)[cj] + (cj // m) * n == prow)[0][0])
        np.testing.assert_array_equal(gotc[r], want[cj * ns + slot])
