"""numpy front-end of libdet6d_oracle.so (see oracle/det6d_oracle.c).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdet6d_oracle.so")

_c_int = ctypes.c_int
_c_float = ctypes.c_float
_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int)
_lp = ctypes.POINTER(ctypes.c_int64)
_up = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    """Compile the oracle with gcc (seconds). Building the checker is not using it."""
    src = os.path.join(_HERE, "det6d_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "det6d_math.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "libdet6d_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _pf(a):
    return a.ctypes.data_as(_fp)


def _pi(a):
    return a.ctypes.data_as(_ip)


def opt_n_threads(n):
    return int(lib().det6d_oracle_opt_n_threads(_c_int(n)))


def set_sqdist_order(alt):
    """0: the shipped contraction fma(dz,dz, fma(dx,dx, dy*dy)); 1: SURVEY.md A.2's fma(dz,dz, fma(dy,dy, dx*dx)) — every
    squared distance of the oracle (FPS, ball queries, 3-NN) follows.  Only tests/test_contraction_order.py uses 1."""
    lib().det6d_oracle_set_sqdist_order(int(bool(alt)))


def fps(xyz, m, temp=None):
    xyz = _f(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, np.float32) if temp is None else temp
    idx = np.zeros((b, m), np.int32)
    rc = lib().det6d_oracle_fps(b, n, m, _pf(xyz), _pf(temp), _pi(idx))
    assert rc == 0
    return idx


def fps_weights(xyz, weights, m, temp=None):
    xyz, weights = _f(xyz), _f(weights)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, np.float32) if temp is None else temp
    idx = np.zeros((b, m), np.int32)
    rc = lib().det6d_oracle_fps_weights(b, n, m, _pf(xyz), _pf(weights), _pf(temp), _pi(idx))
    assert rc == 0
    return idx


def gather_points(points, idx):
    points, idx = _f(points), _i(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.empty((b, c, m), np.float32)
    lib().det6d_oracle_gather_points(b, c, n, m, _pf(points), _pi(idx), _pf(out))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, idx = _f(grad_out), _i(idx)
    b, c, m = grad_out.shape
    gp = np.zeros((b, c, n), np.float32)
    lib().det6d_oracle_gather_points_grad(b, c, n, m, _pf(grad_out), _pi(idx), _pf(gp))
    return gp


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f(xyz), _f(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().det6d_oracle_ball_query(b, n, m, _c_float(radius), nsample, _pf(new_xyz), _pf(xyz), _pi(idx))
    return idx


def ball_query_cnt(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f(xyz), _f(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    cnt = np.zeros((b, m), np.int32)
    lib().det6d_oracle_ball_query_cnt(b, n, m, _c_float(radius), nsample, _pf(new_xyz), _pf(xyz),
                                      _pi(cnt), _pi(idx))
    return cnt, idx


def ball_query_dilated(radius_in, radius_out, nsample, xyz, new_xyz):
    xyz, new_xyz = _f(xyz), _f(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    cnt = np.zeros((b, m), np.int32)
    lib().det6d_oracle_ball_query_dilated(b, n, m, _c_float(radius_in), _c_float(radius_out), nsample,
                                          _pf(new_xyz), _pf(xyz), _pi(cnt), _pi(idx))
    return cnt, idx


def group_points(points, idx):
    points, idx = _f(points), _i(idx)
    b, c, n = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, c, m, ns), np.float32)
    lib().det6d_oracle_group_points(b, c, n, m, ns, _pf(points), _pi(idx), _pf(out))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, idx = _f(grad_out), _i(idx)
    b, c, m, ns = grad_out.shape
    gp = np.zeros((b, c, n), np.float32)
    lib().det6d_oracle_group_points_grad(b, c, n, m, ns, _pf(grad_out), _pi(idx), _pf(gp))
    return gp


def three_nn(unknown, known):
    unknown, known = _f(unknown), _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    lib().det6d_oracle_three_nn(b, n, m, _pf(unknown), _pf(known), _pf(d2), _pi(idx))
    return d2, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f(points), _i(idx), _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.empty((b, c, n), np.float32)
    lib().det6d_oracle_three_interpolate(b, c, m, n, _pf(points), _pi(idx), _pf(weight), _pf(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f(grad_out), _i(idx), _f(weight)
    b, c, n = grad_out.shape
    gp = np.zeros((b, c, m), np.float32)
    lib().det6d_oracle_three_interpolate_grad(b, c, n, m, _pf(grad_out), _pi(idx), _pf(weight), _pf(gp))
    return gp


def boxes_overlap_bev(boxes_a, boxes_b):
    a, b = _f(boxes_a), _f(boxes_b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().det6d_oracle_boxes_overlap_bev(a.shape[0], _pf(a), b.shape[0], _pf(b), _pf(out))
    return out


def boxes_iou_bev(boxes_a, boxes_b):
    a, b = _f(boxes_a), _f(boxes_b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().det6d_oracle_boxes_iou_bev(a.shape[0], _pf(a), b.shape[0], _pf(b), _pf(out))
    return out


def nms_mask(boxes, thresh, normal=False):
    boxes = _f(boxes)
    k = boxes.shape[0]
    cb = (k + 63) // 64
    mask = np.zeros((k, max(cb, 1)), np.uint64)
    lib().det6d_oracle_nms_mask(k, _pf(boxes), _c_float(thresh), int(normal), mask.ctypes.data_as(_up))
    return mask[:, :cb]


def nms(boxes, thresh, normal=False):
    """boxes (K,7) already sorted by descending score -> keep indices (int64)."""
    boxes = _f(boxes)
    k = boxes.shape[0]
    cb = (k + 63) // 64
    mask = np.zeros(k * cb + 1, np.uint64)
    keep = np.zeros(max(k, 1), np.int64)
    num = _c_int(0)
    fn = lib().det6d_oracle_nms_normal if normal else lib().det6d_oracle_nms
    fn(k, _pf(boxes), _c_float(thresh), mask.ctypes.data_as(_up), keep.ctypes.data_as(_lp), ctypes.byref(num))
    return keep[:num.value].copy()


def nms_from_iou(iou, thresh):
    iou = _f(iou)
    k = iou.shape[0]
    keep = np.zeros(max(k, 1), np.int64)
    n = lib().det6d_oracle_nms_from_iou(k, _pf(iou), _c_float(thresh), keep.ctypes.data_as(_lp))
    return keep[:n].copy()


class _LinearArgs(ctypes.Structure):
    _fields_ = [("mode", _c_int), ("rows", _c_int), ("k", _c_int), ("ncols", _c_int),
                ("a", _fp), ("lda", _c_int),
                ("w", _fp), ("ldw", _c_int),
                ("shift", _fp),
                ("act", _c_int),
                ("y", _fp), ("ldy", _c_int), ("col0", _c_int),
                ("n", _c_int), ("m", _c_int), ("ns", _c_int),
                ("idx", _ip),
                ("ctr", _fp), ("ldctr", _c_int),
                ("pool", _c_int),
                ("cnt", _ip),
                ("hdr", _ip), ("crow_p", _ip), ("crow_c", _ip), ("ncols_pad", _c_int)]


def compact_groups(cnt, idx, n, smin=1, split=1):
    """sequential restatement of the compact row lists (csrc/compact.hip): returns hdr, crow_p, crow_c"""
    cnt, idx = _i(cnt), _i(idx)
    b, m, ns = idx.shape
    cap = int(lib().det6d_oracle_compact_rows_capacity(b * m, ns))
    hdr = np.zeros(int(lib().det6d_oracle_compact_hdr_ints(b * m)), np.int32)
    crow_p, crow_c = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    rc = lib().det6d_oracle_compact_groups(b, n, m, ns, smin, split, _pi(cnt), _pi(idx), _pi(hdr), _pi(crow_p), _pi(crow_c),
                                           None, 0, 0, 0)
    assert rc == 0
    return hdr, crow_p, crow_c


def group_expand(p, pcol0, w, shift, act, c1, rows_pts, ctr, ldo, idx=None, lists=None):
    """first layer of a grouped MLP from per-point partial sums (csrc/expand.hip, restated literally): dense rows
    (idx (B,m,ns)) or a compact list (hdr, crow_p, crow_c); returns (rows, ldo)"""
    p, w, shift, pts, ctr = _f(p), _f(w), _f(shift), _f(rows_pts), _f(ctr)
    if lists is not None:
        hdr, crow_p, crow_c = lists
        rows = len(crow_p)
        out = np.full((rows, ldo), -3.0, np.float32)
        rc = lib().det6d_oracle_group_expand(rows, c1, _pf(p), p.shape[-1], pcol0, _pf(w), w.shape[1], _pf(shift), act, _pf(pts),
                                             pts.shape[-1], _pf(ctr), ctr.shape[-1], None, 0, 0, 0, _pi(hdr), _pi(crow_p), _pi(crow_c),
                                             _pf(out), ldo)
    else:
        idx = _i(idx)
        b, m, ns = idx.shape
        rows = b * m * ns
        out = np.full((rows, ldo), -3.0, np.float32)
        rc = lib().det6d_oracle_group_expand(rows, c1, _pf(p), p.shape[-1], pcol0, _pf(w), w.shape[1], _pf(shift), act, _pf(pts),
                                             pts.shape[-1], _pf(ctr), ctr.shape[-1], _pi(idx), pts.shape[1], m, ns, None, None, None,
                                             _pf(out), ldo)
    assert rc == 0
    return out


def mlp_chain3_compact(rows_pts, lists, ctr, layers, out, col0=0):
    """three pointwise layers + max over every centre's rows on a compact list (hdr, crow_p, crow_c);
    layers = [(W, shift)] x 3 with the true widths; out must be zeroed where centres have several parts"""
    hdr, crow_p, crow_c = lists
    a, ctr = _f(rows_pts), _f(ctr)
    ws = [(_f(w), _f(s)) for w, s in layers]
    assert out.dtype == np.float32 and out.flags.c_contiguous
    rc = lib().det6d_oracle_mlp_chain3_compact(
        len(crow_p), _pi(hdr), _pi(crow_p), _pi(crow_c), _pf(a), a.shape[-1], _pf(ctr), ctr.shape[-1],
        _pf(ws[0][0]), ws[0][0].shape[1], _pf(ws[0][1]), len(ws[0][1]), _pf(ws[1][0]), ws[1][0].shape[1], _pf(ws[1][1]),
        len(ws[1][1]), _pf(ws[2][0]), ws[2][0].shape[1], _pf(ws[2][1]), len(ws[2][1]), _pf(out), out.shape[-1], col0)
    assert rc == 0
    return out


def linear(a, w, shift=None, act=0, k=None, idx=None, ctr=None, cnt=None, pool=0, out=None, col0=0):
    """Y = act(A' W + shift).  a: (R,lda) rows, or with idx (B,m,ns): point rows (B,n,lda) gathered
    and centre-subtracted on the first 3 columns (ctr (B,m,ldctr)).  pool=ns -> masked max-pool."""
    a, w = _f(a), _f(w)
    kk = w.shape[0] if k is None else k
    ncols = w.shape[1]
    g = _LinearArgs()
    keep = [a, w]
    if idx is not None:
        idx, ctr = _i(idx), _f(ctr)
        bsz, m, ns = idx.shape
        g.mode, g.rows = 1, bsz * m * ns
        g.n, g.m, g.ns = a.shape[1], m, ns
        g.idx, g.ctr, g.ldctr = _pi(idx), _pf(ctr), ctr.shape[-1]
        g.lda = a.shape[-1]
        keep += [idx, ctr]
    else:
        a2 = a.reshape(-1, a.shape[-1])
        g.mode, g.rows, g.lda = 0, a2.shape[0], a2.shape[-1]
    g.k, g.ncols = kk, ncols
    g.a, g.w, g.ldw = _pf(a), _pf(w), w.shape[1]
    if shift is not None:
        shift = _f(shift)
        keep.append(shift)
        g.shift = _pf(shift)
    g.act = act
    g.pool = pool
    nrows_out = g.rows // pool if pool else g.rows
    if cnt is not None:
        cnt = _i(cnt)
        keep.append(cnt)
        g.cnt = _pi(cnt)
    if out is None:
        out = np.zeros((nrows_out, ncols), np.float32)
    assert out.dtype == np.float32 and out.flags.c_contiguous
    g.y, g.ldy, g.col0 = _pf(out), out.shape[-1], col0
    rc = lib().det6d_oracle_linear(ctypes.byref(g))
    assert rc == 0
    return out


def group_maxpool(x, ns, ncols, cnt=None):
    """(groups * ns, ldx) rows -> (groups, ncols): mask by cnt > 0, then max over the ns rows of a group"""
    x = _f(x)
    groups = x.shape[0] // ns
    y = np.zeros((groups, ncols), np.float32)
    c = _i(cnt).reshape(-1) if cnt is not None else None
    rc = lib().det6d_oracle_group_maxpool(groups, ns, ncols, _pf(x), x.shape[1], _pi(c) if c is not None else None, _pf(y),
                                          ncols, 0)
    assert rc == 0
    return y


def sigmoid_pow(scores, gamma=1.0):
    s = _f(scores)
    out = np.empty_like(s)
    lib().det6d_oracle_sigmoid_pow(s.size, _pf(s), _c_float(gamma), _pf(out))
    return out


def pack_points(points, ld):
    p = _f(points)
    total, width = p.shape
    cin = width - 4
    rows = np.empty((total, ld), np.float32)
    xyz = np.empty((total, 3), np.float32)
    lib().det6d_oracle_pack_points(total, cin, _pf(p), ld, _pf(rows), _pf(xyz))
    return rows, xyz


def gather_rows(rows_in, idx, ncol, ld_out=None):
    rows_in, idx = _f(rows_in), _i(idx)
    b, n, ld_in = rows_in.shape
    m = idx.shape[1]
    ld_out = ncol if ld_out is None else ld_out
    out = np.zeros((b, m, ld_out), np.float32)
    lib().det6d_oracle_gather_rows(b, n, m, ld_in, ld_out, ncol, _pf(rows_in), _pi(idx), _pf(out))
    return out


def vote_points(off, cand, rng):
    off, cand = _f(off), _f(cand)
    rows = off.shape[0]
    vote = np.zeros((rows, 3), np.float32)
    off_out = np.zeros((rows, 3), np.float32)
    lib().det6d_oracle_vote_points(rows, _pf(off), off.shape[1], _pf(cand), cand.shape[1],
                                   _c_float(rng[0]), _c_float(rng[1]), _c_float(rng[2]),
                                   _pf(vote), 3, _pf(off_out))
    return vote, off_out


def decode_boxes(code, pts, nbin=12, ground_aware=True, minus=False, threshold_deg=10.0, factor_deg=45.0):
    code, pts = _f(code), _f(pts)
    rows = code.shape[0]
    boxes = np.empty((rows, 9), np.float32)
    thr = np.float32(np.deg2rad(threshold_deg))
    fac = np.float32(np.deg2rad(factor_deg))
    lib().det6d_oracle_decode_boxes(rows, nbin, int(ground_aware), int(minus), _c_float(thr), _c_float(fac),
                                    _pf(code), code.shape[1], _pf(pts), pts.shape[1], _pf(boxes))
    return boxes


def postprocess(cls, boxes, b, score_thr, pre_max, post_max, nms_thr):
    cls, boxes = _f(cls), _f(boxes)
    p = cls.shape[0] // b
    ncls = cls.shape[1]
    ob = np.zeros((b, post_max, 9), np.float32)
    os_ = np.zeros((b, post_max), np.float32)
    ol = np.zeros((b, post_max), np.int32)
    oi = np.zeros((b, post_max), np.int32)
    oc = np.zeros((b,), np.int32)
    lib().det6d_oracle_postprocess(b, p, ncls, _pf(cls), _pf(boxes), _c_float(score_thr), pre_max, post_max,
                                   _c_float(nms_thr), _pf(ob), _pf(os_), _pi(ol), _pi(oi), _pi(oc))
    return ob, os_, ol, oi, oc


def math_fn(name, x, y=None):
    fn = {"exp": 0, "log": 1, "sin": 2, "cos": 3, "atan2": 4, "sigmoid": 5}[name]
    x = _f(x)
    y = _f(y) if y is not None else x
    out = np.empty_like(x)
    lib().det6d_oracle_math(fn, x.size, _pf(x), _pf(y), _pf(out))
    return out


def fps_fused(xyz, lo, hi, m, scores, gamma, idx_out, idx_offset):
    xyz = _f(xyz)
    b, n_total, _ = xyz.shape
    sc = _f(scores) if scores is not None else None
    assert idx_out.dtype == np.int32 and idx_out.flags.c_contiguous
    rc = lib().det6d_oracle_fps_fused(b, n_total, lo, hi, m, _pf(xyz), _pf(sc) if sc is not None else None,
                                      _c_float(gamma), None, _pi(idx_out), idx_out.shape[1], idx_offset)
    assert rc == 0
    return idx_out


def gather_centres(xyz, idx, ld_rows=0, zero_from=0):
    xyz, idx = _f(xyz), _i(idx)
    b, n, _ = xyz.shape
    m = idx.shape[1]
    out = np.empty((b, m, 3), np.float32)
    rows = np.full((b, m, ld_rows), 7.0, np.float32) if ld_rows else None
    lib().det6d_oracle_gather_centres(b, n, m, _pf(xyz), _pi(idx), _pf(out), _pf(rows) if rows is not None else None,
                                      ld_rows, zero_from)
    return out, rows


def with_batch_index(src, ncol=3):
    src = _f(src)
    b, m, ld = src.shape
    dst = np.empty((b * m, ncol + 1), np.float32)
    lib().det6d_oracle_with_batch_index(b, m, _pf(src), ld, ncol, _pf(dst))
    return dst


def prepare_points(frames, point_cloud_range, num_points, seed, near_depth=40.0, scene_ids=None):
    """frames: list of (n_i, C) raw frames -> (points (B*N, 1+C), n_in_range (B))"""
    import ctypes
    frames = [_f(f) for f in frames]
    b, c = len(frames), frames[0].shape[1]
    offsets = np.zeros(b + 1, np.int32)
    offsets[1:] = np.cumsum([f.shape[0] for f in frames])
    raw = np.ascontiguousarray(np.concatenate(frames, axis=0)) if offsets[-1] else np.zeros((1, c), np.float32)
    out = np.empty((b * num_points, 1 + c), np.float32)
    n_in = np.zeros(b, np.int32)
    r = [float(v) for v in point_cloud_range]
    ids = None if scene_ids is None else _i(scene_ids)
    lib().det6d_oracle_prepare_points(b, _pi(offsets), None if ids is None else _pi(ids), c, _pf(raw), _c_float(r[0]), _c_float(r[1]), _c_float(r[3]),
                                      _c_float(r[4]), int(num_points), _c_float(near_depth), ctypes.c_uint64(seed),
                                      _pf(out), _pi(n_in))
    return out, n_in


def perm(n, seed, scene=0, purpose=2):
    import ctypes
    out = np.empty(n, np.uint32)
    lib().det6d_oracle_perm(ctypes.c_uint32(n), ctypes.c_uint64(seed), ctypes.c_uint32(scene), ctypes.c_uint32(purpose),
                            out.ctypes.data_as(ctypes.c_void_p))
    return out


def kitti_annos(boxes, scene_of, calib):
    """C-oracle mirror of det6d_kitti_annos: boxes (T, ld), scene_of (T) int32, calib (B, 28) -> (T, 12)"""
    boxes, calib, scene_of = _f(boxes), _f(calib), _i(scene_of)
    out = np.empty((boxes.shape[0], 12), np.float32)
    lib().det6d_oracle_kitti_annos(boxes.shape[0], _pf(boxes), boxes.shape[1], _pi(scene_of), _pf(calib), _pf(out))
    return out


def make_slope(points, boxes9, params):
    """C-oracle mirror of det6d_make_slope; returns updated copies"""
    pts = np.array(points, np.float32, copy=True, order='C')
    bx = np.array(boxes9, np.float64, copy=True, order='C')
    prm = np.ascontiguousarray(params, np.float64)
    dp = ctypes.c_void_p
    lib().det6d_oracle_make_slope(pts.shape[0], pts.ctypes.data_as(dp), pts.shape[1], bx.shape[0], bx.ctypes.data_as(dp),
                                  prm.ctypes.data_as(dp))
    return pts, bx


def boxes9_corners(boxes9):
    bx = np.ascontiguousarray(boxes9, np.float64)
    out = np.empty((bx.shape[0], 8, 3), np.float64)
    lib().det6d_oracle_boxes9_corners(bx.shape[0], bx.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


# ---------------------------------------------------------------- KITTI evaluator (oracle backend)
class _EvalMatchArgs(ctypes.Structure):
    _fields_ = [("n_frames", ctypes.c_int), ("n_thresh", ctypes.c_int), ("metric", ctypes.c_int),
                ("compute_aos", ctypes.c_int), ("dt_f32", ctypes.c_int), ("min_overlap", ctypes.c_double)] + \
               [(n, ctypes.c_void_p) for n in ("thresholds", "dt_off", "gt_off", "dc_off", "pair_off", "overlaps", "gt_alpha",
                                                "dt_bbox", "dt_alpha", "dt_score", "ignored_gt", "ignored_dt", "dc_bbox",
                                                "workspace", "stats", "tp_scores", "tp_count", "gt_of_tp")]


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class EvalBackend(object):
    """CPU-oracle counterpart of de6d_amd.ops.kitti_eval.DeviceEvalBackend (same methods, NumPy arrays)"""

    def __init__(self, layout):
        self.lay = layout
        c = np.ascontiguousarray
        self.dt_off, self.gt_off = c(layout.dt_off, np.int32), c(layout.gt_off, np.int32)
        self.pair_off = c(layout.pair_off, np.int64)
        self.gt_alpha, self.dt_alpha = c(layout.gt_alpha, np.float64), c(layout.dt_alpha, np.float64)
        self.dt_score, self.dt_bbox = c(layout.dt_score, np.float64), c(layout.dt_boxes[0] if 0 in layout.dt_boxes else np.zeros((1, 4)), np.float64)
        self._ov = {}

    def overlaps(self, metric):
        if metric not in self._ov:
            n_pairs = int(self.pair_off[-1])
            out = np.zeros(max(n_pairs, 1), np.float64)
            dt, gt = (np.ascontiguousarray(b[metric], np.float64) for b in (self.lay.dt_boxes, self.lay.gt_boxes))
            lib().det6d_oracle_eval_overlaps(metric, self.lay.n_frames, _vp(self.dt_off), _vp(self.gt_off), _vp(self.pair_off),
                                             ctypes.c_int64(n_pairs), _vp(dt), _vp(gt), int(self.lay.dt_f32), _vp(out))
            self._ov[metric] = out
        return self._ov[metric]

    def overlaps_host(self, metric):
        return self.overlaps(metric)[:int(self.pair_off[-1])]

    def _args(self, metric, ignored_gt, ignored_dt, dc_off, dc_bbox, min_overlap, n_thresh, compute_aos):
        keep = [np.ascontiguousarray(ignored_gt, np.int32), np.ascontiguousarray(ignored_dt, np.int32),
                np.ascontiguousarray(dc_off, np.int32),
                np.ascontiguousarray(dc_bbox if len(dc_bbox) else np.zeros((1, 4)), np.float64), self.overlaps(metric)]
        a = _EvalMatchArgs()
        a.n_frames, a.n_thresh, a.metric, a.compute_aos, a.dt_f32 = self.lay.n_frames, n_thresh, metric, int(compute_aos), int(self.lay.dt_f32)
        a.min_overlap = float(min_overlap)
        for name, arr in (("dt_off", self.dt_off), ("gt_off", self.gt_off), ("dc_off", keep[2]), ("pair_off", self.pair_off),
                          ("overlaps", keep[4]), ("gt_alpha", self.gt_alpha), ("dt_bbox", self.dt_bbox), ("dt_alpha", self.dt_alpha),
                          ("dt_score", self.dt_score), ("ignored_gt", keep[0]), ("ignored_dt", keep[1]), ("dc_bbox", keep[3])):
            setattr(a, name, arr.ctypes.data)
        return a, keep

    def pass_a(self, metric, ignored_gt, ignored_dt, dc_off, dc_bbox, min_overlap, want_gt_of_tp=False):
        a, keep = self._args(metric, ignored_gt, ignored_dt, dc_off, dc_bbox, min_overlap, 0, False)
        tp_scores = np.zeros(max(int(self.gt_off[-1]), 1), np.float64)
        tp_count = np.zeros(max(self.lay.n_frames, 1), np.int32)
        gt_of_tp = np.full(max(int(self.dt_off[-1]), 1), -1, np.int32)
        a.tp_scores, a.tp_count = tp_scores.ctypes.data, tp_count.ctypes.data
        a.gt_of_tp = gt_of_tp.ctypes.data if want_gt_of_tp else None
        lib().det6d_oracle_eval_match(ctypes.byref(a))
        return tp_scores[:int(self.gt_off[-1])], tp_count[:self.lay.n_frames], gt_of_tp[:int(self.dt_off[-1])] if want_gt_of_tp else None

    def pass_b(self, metric, ignored_gt, ignored_dt, dc_off, dc_bbox, min_overlap, thresholds, compute_aos):
        n_thresh = len(thresholds)
        if n_thresh == 0:
            return np.zeros((0, 4))
        a, keep = self._args(metric, ignored_gt, ignored_dt, dc_off, dc_bbox, min_overlap, n_thresh, compute_aos)
        thr = np.ascontiguousarray(thresholds, np.float64)
        stats = np.zeros(max(self.lay.n_frames, 1) * n_thresh * 4, np.float64)
        pr = np.zeros((n_thresh, 4), np.float64)
        a.thresholds, a.stats = thr.ctypes.data, stats.ctypes.data
        lib().det6d_oracle_eval_match(ctypes.byref(a))
        lib().det6d_oracle_eval_reduce(self.lay.n_frames, n_thresh, _vp(stats), _vp(pr))
        return pr
