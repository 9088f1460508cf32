"""Compact (ragged) row lists of the grouped SA MLPs (csrc/compact.hip): the reference pads a ball holding
cnt < nsample points with repetitions of its first cnt hits (ball_query_gpu.cu:75-90,114-129), so evaluating
only the first 2^ceil(log2 cnt) slots of a centre must give the DENSE oracle's pooled features bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def padded_query(rng, b, n, m, ns, p_empty=0.05, p_full=0.1):
    """(cnt, idx) with the reference's layout: ascending hits, then periodic repetition; zeros when empty"""
    cnt = np.minimum(rng.geometric(0.25, (b, m)), ns).astype(np.int32)
    u = rng.uniform(size=(b, m))
    cnt[u < p_empty] = 0
    cnt[u > 1 - p_full] = ns
    idx = np.zeros((b, m, ns), np.int32)
    for bi in range(b):
        for j in range(m):
            c = cnt[bi, j]
            if c:
                hits = np.sort(rng.choice(n, c, replace=False))
                idx[bi, j] = hits[np.arange(ns) % c]
    return cnt, idx


def build_list(fused, cnt, idx, n, smin, split):
    old = fused.COMPACT_SMIN, fused.COMPACT_SPLIT
    fused.COMPACT_SMIN, fused.COMPACT_SPLIT = smin, split
    try:
        return fused.compact_groups(dev(cnt), dev(idx), n)
    finally:
        fused.COMPACT_SMIN, fused.COMPACT_SPLIT = old


@pytest.mark.parametrize("b,n,m,ns,smin", [(2, 500, 300, 16, 4), (3, 900, 1000, 32, 4), (1, 64, 7, 32, 1), (2, 300, 513, 16, 2),
                                           (8, 2048, 4096, 32, 4), (2, 100, 50, 8, 4), (1, 40, 33, 4, 4)])
def test_split_groups_structure(b, n, m, ns, smin):
    """split lists (granule 4): up to 4 hits one part of the next power of two >= smin, beyond that ceil(cnt / 4) * 4
    rows cut along their binary digits into parts of descending size; part of size S at slot offset
    (rows & ~(2S - 1)) lies in the class-S region, centres ascending"""
    from de6d_amd.ops import fused
    rng = np.random.default_rng(b * 1000 + m + ns + 5)
    cnt, idx = padded_query(rng, b, n, m, ns)
    gran = min(4, ns)
    cr = build_list(fused, cnt, idx, n, smin, gran)
    hdr, cp, cc = cr.hdr.cpu().numpy(), cr.crow_p.cpu().numpy(), cr.crow_c.cpu().numpy()
    flat_cnt, flat_idx = cnt.reshape(-1), idx.reshape(-1, ns)
    kk = np.clip(flat_cnt, 1, ns)
    pow2 = np.maximum(smin, 2 ** np.ceil(np.log2(kk)).astype(np.int64))
    rows = np.where(kk > gran, (kk + gran - 1) // gran * gran, pow2)
    assert hdr[0] % 128 == 0 and hdr[7] == b * m and hdr[8] == np.minimum(cnt, ns).sum() and hdr[9] == rows.sum()
    start = 0
    for c in range(6):
        s, end = 32 >> c, int(hdr[1 + c])
        members = np.nonzero(rows & s)[0]
        if len(members) == 0:
            assert end == start
            continue
        assert end % 128 == 0 and start + len(members) * s <= end < start + len(members) * s + 128
        off = rows[members] & ~(2 * s - 1)
        region_c = cc[start:start + len(members) * s].reshape(-1, s)
        region_p = cp[start:start + len(members) * s].reshape(-1, s)
        np.testing.assert_array_equal(region_c & 0x1fffffff, np.repeat(members[:, None], s, 1))
        np.testing.assert_array_equal((region_c >> 30) & 1, np.repeat((flat_cnt[members] == 0)[:, None], s, 1))
        multi = (rows[members] & (rows[members] - 1)) != 0
        np.testing.assert_array_equal((region_c >> 29) & 1, np.repeat(multi[:, None], s, 1))
        want_p = np.stack([flat_idx[mm, o:o + s] for mm, o in zip(members, off)]) + (members // m * n)[:, None]
        np.testing.assert_array_equal(region_p, want_p)
        assert (cc[start + len(members) * s:end] == -1).all()
        start = end
    assert hdr[0] == start


@pytest.mark.parametrize("b,n,m,ns,smin", [(2, 500, 300, 16, 4), (3, 900, 1000, 32, 4), (1, 64, 7, 32, 1), (2, 300, 513, 16, 2),
                                           (8, 2048, 4096, 32, 4), (2, 100, 50, 8, 4), (1, 40, 33, 4, 4)])
def test_compact_groups_structure(b, n, m, ns, smin):
    from de6d_amd.ops import fused
    rng = np.random.default_rng(b * 1000 + m + ns)
    cnt, idx = padded_query(rng, b, n, m, ns)
    cr = build_list(fused, cnt, idx, n, smin, 0)
    hdr, cp, cc = cr.hdr.cpu().numpy(), cr.crow_p.cpu().numpy(), cr.crow_c.cpu().numpy()
    total = int(hdr[0])
    assert total % 128 == 0 and total <= cr.capacity and cr.capacity % 1024 == 0
    assert hdr[6] == total and hdr[7] == b * m and hdr[8] == np.minimum(cnt, ns).sum()
    want_cls = np.maximum(smin, 2 ** np.ceil(np.log2(np.maximum(cnt, 1))).astype(np.int64)).reshape(-1)
    want_cls = np.minimum(want_cls, ns)
    seen = np.zeros(b * m, np.int64)
    start = 0
    for c in range(6):
        s, end = 32 >> c, int(hdr[1 + c])
        assert end % 128 == 0 and end >= start
        members = np.nonzero(want_cls == s)[0]          # ascending centre order inside a region
        if len(members) == 0:
            assert end == start
            continue
        assert start + len(members) * s <= end < start + len(members) * s + 128
        region_c = cc[start:start + len(members) * s].reshape(-1, s)
        region_p = cp[start:start + len(members) * s].reshape(-1, s)
        flat_cnt = cnt.reshape(-1)[members]
        np.testing.assert_array_equal(region_c & 0x3fffffff, np.repeat(members[:, None], s, 1))
        np.testing.assert_array_equal((region_c >> 30) & 1, np.repeat((flat_cnt == 0)[:, None], s, 1))
        np.testing.assert_array_equal(region_p, idx.reshape(-1, ns)[members][:, :s] + (members // m * n)[:, None])
        assert (cc[start + len(members) * s:end] == -1).all()
        assert ((cp[start + len(members) * s:end] >= 0) & (cp[start + len(members) * s:end] < b * n)).all()
        seen[members] += 1
        start = end
    assert (seen == 1).all()
    assert hdr[9] == (want_cls).sum()


@pytest.mark.parametrize("smin,split", [(1, 1), (1, 4), (4, 4), (4, 0), (2, 8)])
def test_compact_groups_equals_c_oracle(oracle_ops, smin, split):
    """device lists == the sequential C restatement, word for word (live rows; the rest of the capacity is scratch)"""
    from de6d_amd.ops import fused
    rng = np.random.default_rng(17 + smin + split)
    b, n, m, ns = 3, 777, 1500, 32
    cnt, idx = padded_query(rng, b, n, m, ns)
    cr = build_list(fused, cnt, idx, n, smin, split)
    ohdr, op, oc = oracle_ops.compact_groups(cnt, idx, n, smin=smin, split=max(split, smin) if split else 0)
    hdr = cr.hdr.cpu().numpy()
    np.testing.assert_array_equal(hdr[:10], ohdr[:10])
    live = int(hdr[0])
    np.testing.assert_array_equal(cr.crow_p.cpu().numpy()[:live], op[:live])
    np.testing.assert_array_equal(cr.crow_c.cpu().numpy()[:live], oc[:live])


@pytest.mark.parametrize("kind", ["all_empty", "all_full", "all_single", "mixed_heavy_tail", "one_centre", "ragged_total"])
def test_compact_groups_extreme_distributions(oracle_ops, kind):
    """row lists on degenerate hit-count distributions == the C oracle's, and the MLP over them == the dense oracle"""
    from de6d_amd.ops import fused
    rng = np.random.default_rng(sum(map(ord, kind)))
    b, n, ns = 2, 400, 32
    m = {"one_centre": 1, "ragged_total": 257}.get(kind, 300)
    cnt = {"all_empty": np.zeros((b, m)), "all_full": np.full((b, m), ns), "all_single": np.ones((b, m)),
           "mixed_heavy_tail": np.minimum(rng.zipf(1.5, (b, m)), ns), "one_centre": np.array([[5], [0]]),
           "ragged_total": rng.integers(0, ns + 1, (b, m))}[kind].astype(np.int32)
    idx = np.zeros((b, m, ns), np.int32)
    for bi in range(b):
        for j in range(m):
            c = cnt[bi, j]
            if c:
                idx[bi, j] = np.sort(rng.choice(n, c, replace=False))[np.arange(ns) % c]
    for smin, split in ((1, 1), (4, 4)):
        cr = build_list(fused, cnt, idx, n, smin, split)
        ohdr, op, oc = oracle_ops.compact_groups(cnt, idx, n, smin=smin, split=split)
        hdr = cr.hdr.cpu().numpy()
        np.testing.assert_array_equal(hdr[:10], ohdr[:10])
        live = int(hdr[0])
        np.testing.assert_array_equal(cr.crow_p.cpu().numpy()[:live], op[:live])
        np.testing.assert_array_equal(cr.crow_c.cpu().numpy()[:live], oc[:live])
    c_in, widths = 1, (32, 32, 64)
    ld = 4
    rows = np.zeros((b, n, ld), np.float32)
    rows[...] = rng.normal(size=(b, n, ld))
    ctr = rng.normal(size=(b, m, 3)).astype(np.float32)
    layers_np, layers_dev = make_layers(rng, ld, c_in, widths)
    out = torch.zeros((b * m, widths[2]), device="cuda")
    cr = build_list(fused, cnt, idx, n, 1, 1)
    fused.mlp_chain3_compact(dev(rows), cr, dev(ctr), layers_dev, out, 0)
    h = oracle_ops.linear(rows, layers_np[0][0], layers_np[0][1], 1, idx=idx, ctr=ctr)
    h = oracle_ops.linear(h, layers_np[1][0], layers_np[1][1], 1)
    ref = oracle_ops.linear(h, layers_np[2][0], layers_np[2][1], 1, cnt=cnt, pool=ns)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def make_layers(rng, ld, c_in, widths):
    dims = [ld] + list(widths)
    layers_np, layers_dev = [], []
    for i in range(3):
        kin = dims[i] if i == 0 else (dims[i] + 3) // 4 * 4
        wpad = (dims[i + 1] + 3) // 4 * 4
        w = np.zeros((kin, wpad), np.float32)
        k_used = dims[i] if i else 3 + c_in
        w[:k_used, :dims[i + 1]] = rng.normal(size=(k_used, dims[i + 1])) / np.sqrt(dims[i])
        s = rng.normal(size=(dims[i + 1],)).astype(np.float32)
        layers_np.append((w, s))
        layers_dev.append((dev(w), dev(s), dims[i + 1], 1))
    return layers_np, layers_dev


@pytest.mark.parametrize("c_in,widths,ns,chain", [(1, (16, 16, 32), 16, True), (1, (32, 32, 64), 32, True), (64, (64, 64, 128), 16, True),
                                                   (64, (64, 96, 128), 32, True), (1, (32, 32, 64), 32, False),
                                                   (128, (128, 128, 256), 16, False), (128, (128, 256, 256), 32, False),
                                                   (5, (24, 40, 72), 8, False), (256, (256, 512, 1024), 32, False)])
@pytest.mark.parametrize("smin,split", [(4, 4), (4, 0), (1, 0), (1, 4), (1, 1), (2, 2), (2, 8)])
def test_compact_mlp_equals_dense_oracle(oracle_ops, c_in, widths, ns, chain, smin, split):
    """gather + 3 x (GEMM, shift, ReLU) + mask + max-pool over the compact rows (register chain kernels and the
    three-GEMM route) == the oracle over ALL nsample rows"""
    from de6d_amd.ops import fused
    if smin < 4 and not fused.COMPACT_SMALL_CLASSES:
        pytest.skip("row classes below 4 are not enabled in the GEMM epilogues")
    b, n, m = 2, 700, 333
    rng = np.random.default_rng(sum(widths) + ns)
    ld = (3 + c_in + 3) // 4 * 4
    rows = np.zeros((b, n, ld), np.float32)
    rows[..., :3 + c_in] = rng.normal(size=(b, n, 3 + c_in))
    ctr = rng.normal(size=(b, m, 3)).astype(np.float32)
    cnt, idx = padded_query(rng, b, n, m, ns)
    layers_np, layers_dev = make_layers(rng, ld, c_in, widths)
    out = torch.full((b * m, widths[2] + 3), -7.0, device="cuda")
    out[:, 3:] = 0.0        # split lists combine the parts of a centre by an atomic max into a zeroed buffer
    cr = build_list(fused, cnt, idx, n, smin, split)
    d_rows, d_ctr = dev(rows), dev(ctr)
    if chain:
        assert fused.chain_compact_eligible(ld, layers_dev)
        fused.mlp_chain3_compact(d_rows, cr, d_ctr, layers_dev, out, 3)
    else:
        x = None
        for li, (w, s, cout, act) in enumerate(layers_dev):
            if li == 2:
                tgt, kw = out, dict(ncols=cout, col0=3, cnt=dev(cnt), pool=-1)
            else:
                tgt, kw = torch.zeros((cr.capacity, w.shape[1]), device="cuda"), dict(ncols=cout)
            if li == 0:
                fused.linear(d_rows, w, s, act, tgt, ctr=d_ctr, compact=cr, gather=True, **kw)
            else:
                fused.linear(x, w, s, act, tgt, compact=cr, **kw)
            x = tgt
    h = oracle_ops.linear(rows, layers_np[0][0], layers_np[0][1], 1, idx=idx, ctr=ctr)
    h = oracle_ops.linear(np.ascontiguousarray(np.pad(h, ((0, 0), (0, layers_np[1][0].shape[0] - h.shape[1])))), layers_np[1][0], layers_np[1][1], 1)
    h = np.ascontiguousarray(np.pad(h[:, :widths[1]], ((0, 0), (0, layers_np[2][0].shape[0] - widths[1]))))
    ref = np.full((b * m, widths[2] + 3), -7.0, np.float32)
    oracle_ops.linear(h, layers_np[2][0][:, :widths[2]], layers_np[2][1], 1, cnt=cnt, pool=ns, out=ref, col0=3)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("env,select", [
    ({'DET6D_DENSE_ROWS': '1'}, 'tiny or sloped or three_class'), ({'DET6D_COMPACT_NO_CHAIN': '1'}, 'tiny or sloped or three_class'),
    ({'DET6D_COMPACT_SPLIT': '0'}, 'tiny or sloped or three_class'), ({'DET6D_COMPACT_SMIN': '4'}, 'tiny or sloped or three_class'),
    ({'DET6D_COMPACT_SPLIT': '1'}, 'tiny or sloped or three_class'),
    # the round-1 routes behind the round-2 kernels: gathered first-layer GEMM, three launches per wide group, one launch per
    # plain layer (alone and on dense rows), and only the first sampler hoisted out of the captured passes
    ({'DET6D_NO_EXPAND': '1'}, 'sloped or three_class'), ({'DET6D_NO_GROUP_KERNEL': '1'}, 'sloped or three_class'),
    ({'DET6D_NO_ROWS_KERNEL': '1'}, 'tiny or sloped'), ({'DET6D_NO_GROUP_KERNEL': '1', 'DET6D_DENSE_ROWS': '1'}, 'sloped'),
    ({'DET6D_NO_EXPAND': '1', 'DET6D_DENSE_ROWS': '1'}, 'sloped'),
    ({'DET6D_NO_HOIST': '1'}, 'pass_group or two_stage or graph_replay')])
def test_model_parity_on_the_other_row_paths(env, select):
    """the whole-model bit-exact tests run on compact rows with the fused kernels by default; rerun subsets on the
    reference's dense row space (DET6D_DENSE_ROWS: a switch of the shipped library) and on every alternative route of the
    experiments build (DET6D_EXPERIMENTS_LIB=1; switches are read at import: child process)"""
    if set(env) != {'DET6D_DENSE_ROWS'}:
        env = dict(env, DET6D_EXPERIMENTS_LIB='1')
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_model_gpu.py'), '-q', '-x', '-m', 'gpu',
                          '-k', select], env=dict(os.environ, **env), cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'passed' in out.stdout


@pytest.mark.parametrize("c_in,widths,ns", [(128, (128, 128, 256), 16), (128, (128, 256, 256), 32), (256, (256, 256, 512), 16),
                                            (256, (256, 512, 1024), 32)])
@pytest.mark.parametrize("smin,split", [(1, 1), (4, 4), (1, 0), (2, 8)])
def test_group_kernel_equals_dense_oracle(oracle_ops, c_in, widths, ns, smin, split):
    """csrc/mlp_group.hip: per-point first-layer sums + ONE launch for layers 2, 3 and the pooling, over compact lists of
    every granularity and over the dense rows, == the oracle's three layers over ALL nsample rows (bit for bit)"""
    from de6d_amd.ops import fused
    b, n, m = 2, 700, 334
    rng = np.random.default_rng(sum(widths) + ns + smin)
    ld = (3 + c_in + 3) // 4 * 4
    rows = np.zeros((b, n, ld), np.float32)
    rows[..., :3 + c_in] = rng.normal(size=(b, n, 3 + c_in))
    ctr = rng.normal(size=(b, m, 3)).astype(np.float32)
    cnt, idx = padded_query(rng, b, n, m, ns)
    layers_np, layers_dev = make_layers(rng, ld, c_in, widths)
    h = oracle_ops.linear(rows, layers_np[0][0], layers_np[0][1], 1, idx=idx, ctr=ctr)
    h = oracle_ops.linear(h, layers_np[1][0], layers_np[1][1], 1)
    ref = np.full((b * m, widths[2] + 4), -7.0, np.float32)
    oracle_ops.linear(h, layers_np[2][0][:, :widths[2]], layers_np[2][1], 1, cnt=cnt, pool=ns, out=ref, col0=4)
    d_rows, d_ctr = dev(rows), dev(ctr)
    wz = layers_dev[0][0].clone()
    wz[:3] = 0
    p = torch.empty((b * n, wz.shape[1] + 8), device="cuda")            # this group's sums at column 8 of a wider P
    fused.linear(d_rows.view(b * n, ld), wz, None, 0, p, col0=8)
    assert fused.group_kernel_eligible(layers_dev, ns, True) and fused.group_kernel_eligible(layers_dev, ns, False)
    # compact list
    out = torch.full((b * m, widths[2] + 4), -7.0, device="cuda")
    out[:, 4:] = 0.0
    cr = build_list(fused, cnt, idx, n, smin, split)
    fused.mlp_group3(p, 8, layers_dev, d_rows, d_ctr, out, 4, compact=cr)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # dense rows
    out2 = torch.full((b * m, widths[2] + 4), -7.0, device="cuda")
    fused.mlp_group3(p, 8, layers_dev, d_rows, d_ctr, out2, 4, idx=dev(idx), cnt=dev(cnt))
    np.testing.assert_array_equal(out2.cpu().numpy(), ref)


@pytest.mark.parametrize("rb", ["1", "3"])
def test_plain_layer_stacks_other_tiles(rb):
    """DET6D_ROWS_RB: 64-row tiles for the narrow stacks (3 = whatever the row count, so that the small test shapes take
    them; 1 = never).  Same bits (child process: the switch is read once)."""
    if os.environ.get('DET6D_ROWS_RB') is not None:
        pytest.skip('already a child')
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-m', 'gpu', '-k',
                          'test_plain_layer_stacks_equal_the_layer_by_layer_oracle'],
                         env=dict(os.environ, DET6D_ROWS_RB=rb, DET6D_EXPERIMENTS_LIB='1'), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'passed' in out.stdout


@pytest.mark.parametrize("env", [{'DET6D_GROUP_STREAM': '0'}, {'DET6D_GROUP_STREAM': '3'}, {'DET6D_GROUP_PRE': '0'}])
def test_group_kernel_other_forms(env):
    """DET6D_GROUP_STREAM (bit mask, default 2): streaming form (second layer in 128-column chunks, two workgroups per CU) for
    the head's [256 -> 512 -> 1024] group (1) and / or its [256 -> 256 -> 512] group (2); 0 = one-pass form everywhere.
    DET6D_GROUP_PRE=0: the rows' list entries loaded inside the first layer instead of a K loop ahead.  Same bits on every
    route (the switches are read once per process: child process)."""
    if any(k in os.environ for k in env):
        pytest.skip('already a child')
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-m', 'gpu', '-k',
                          'test_group_kernel_equals_dense_oracle or test_group_kernel_on_degenerate_lists'],
                         env=dict(os.environ, DET6D_EXPERIMENTS_LIB='1', **env), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'passed' in out.stdout


@pytest.mark.parametrize("case", ["all_empty", "all_full", "all_single", "one_centre"])
def test_group_kernel_on_degenerate_lists(oracle_ops, case):
    from de6d_amd.ops import fused
    c_in, widths, ns = 128, (128, 256, 256), 32
    b, n, m = 2, 300, 64 if case != "one_centre" else 1
    rng = np.random.default_rng(5)
    ld = (3 + c_in + 3) // 4 * 4
    rows = np.zeros((b, n, ld), np.float32)
    rows[..., :3 + c_in] = rng.normal(size=(b, n, 3 + c_in))
    ctr = rng.normal(size=(b, m, 3)).astype(np.float32)
    cnt = {"all_empty": np.zeros((b, m)), "all_full": np.full((b, m), ns), "all_single": np.ones((b, m)),
           "one_centre": np.full((b, m), 21)}[case].astype(np.int32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    for bi in range(b):
        for j in range(m):
            c = max(int(cnt[bi, j]), 1)
            if cnt[bi, j] == 0:
                idx[bi, j] = 0
            idx[bi, j] = np.resize(idx[bi, j, :c], ns)
    layers_np, layers_dev = make_layers(rng, ld, c_in, widths)
    h = oracle_ops.linear(rows, layers_np[0][0], layers_np[0][1], 1, idx=idx, ctr=ctr)
    h = oracle_ops.linear(h, layers_np[1][0], layers_np[1][1], 1)
    ref = oracle_ops.linear(h, layers_np[2][0][:, :widths[2]], layers_np[2][1], 1, cnt=cnt, pool=ns)
    wz = layers_dev[0][0].clone()
    wz[:3] = 0
    p = torch.empty((b * n, wz.shape[1]), device="cuda")
    fused.linear(dev(rows).view(b * n, ld), wz, None, 0, p)
    out = torch.zeros((b * m, widths[2]), device="cuda")
    cr = build_list(fused, cnt, idx, n, 1, 1)
    fused.mlp_group3(p, 0, layers_dev, dev(rows), dev(ctr), out, 0, compact=cr)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("rows_n", [1, 33, 2047, 4096])
def test_plain_layer_stacks_equal_the_layer_by_layer_oracle(oracle_ops, rows_n):
    """csrc/mlp_rows.hip: aggregation + confidence chain (a hidden layer stored AND fed on), a two-chain launch over one
    input (cls / reg towers), ragged row counts, N = 1 / 3 / 32 outputs — == oracle linear, layer by layer"""
    from de6d_amd.ops import fused
    rng = np.random.default_rng(rows_n)

    def layer(k_rows, k_used, k0, n_out, act):
        w = np.zeros((k_rows, (n_out + 3) // 4 * 4), np.float32)
        w[k0:k0 + k_used, :n_out] = rng.normal(size=(k_used, n_out)) / np.sqrt(k_used)
        return w, rng.normal(size=(n_out,)).astype(np.float32), n_out, act
    # aggregation 96 -> 64 (stored at column 3 of 68-wide rows), confidence 64 -> 32 -> 1 reading weight rows 3..66
    x = rng.normal(size=(rows_n, 96)).astype(np.float32)
    agg, c1, c2 = layer(96, 96, 0, 64, 1), layer(68, 64, 3, 32, 1), layer(32, 32, 0, 1, 0)
    new_rows = torch.full((rows_n, 68), 9.0, device="cuda")
    scores = torch.empty((rows_n, 1), device="cuda")
    spec = [(dev(agg[0]), 0, dev(agg[1]), 96, 64, 1, new_rows, 3), (dev(c1[0]), 3, dev(c1[1]), 64, 32, 1, None, 0),
            (dev(c2[0]), 0, dev(c2[1]), 32, 1, 0, scores, 0)]
    assert fused.mlp_rows_eligible(96, [spec])
    fused.mlp_rows(dev(x), 0, [spec])
    a = oracle_ops.linear(x, agg[0][:, :64], agg[1], 1)
    feat_rows = np.zeros((rows_n, 68), np.float32)
    feat_rows[:, 3:67] = a
    h = oracle_ops.linear(feat_rows, c1[0][:, :32], c1[1], 1)
    sc = oracle_ops.linear(h, c2[0][:, :1], c2[1], 0)
    got = new_rows.cpu().numpy()
    np.testing.assert_array_equal(got[:, 3:67], a)
    assert (got[:, :3] == 9.0).all() and (got[:, 67] == 9.0).all()
    np.testing.assert_array_equal(scores.cpu().numpy(), sc)
    # two towers over one 512-wide input: 512 -> 128 -> 3 and 512 -> 128 -> 32
    x2 = rng.normal(size=(rows_n, 512)).astype(np.float32)
    t1a, t1b, t2a, t2b = layer(512, 512, 0, 128, 1), layer(128, 128, 0, 3, 0), layer(512, 512, 0, 128, 1), layer(128, 128, 0, 32, 0)
    o1 = torch.empty((rows_n, 3), device="cuda")
    o2 = torch.empty((rows_n, 32), device="cuda")
    towers = [[(dev(t1a[0]), 0, dev(t1a[1]), 512, 128, 1, None, 0), (dev(t1b[0]), 0, dev(t1b[1]), 128, 3, 0, o1, 0)],
              [(dev(t2a[0]), 0, dev(t2a[1]), 512, 128, 1, None, 0), (dev(t2b[0]), 0, dev(t2b[1]), 128, 32, 0, o2, 0)]]
    assert fused.mlp_rows_eligible(512, towers)
    fused.mlp_rows(dev(x2), 0, towers)
    r1 = oracle_ops.linear(oracle_ops.linear(x2, t1a[0][:, :128], t1a[1], 1), t1b[0][:, :3], t1b[1], 0)
    r2 = oracle_ops.linear(oracle_ops.linear(x2, t2a[0][:, :128], t2a[1], 1), t2b[0][:, :32], t2b[1], 0)
    np.testing.assert_array_equal(o1.cpu().numpy(), r1)
    np.testing.assert_array_equal(o2.cpu().numpy(), r2)
