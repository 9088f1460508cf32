"""Tensor-level wrappers of the fused engine entry points of libdet6d_hip.so
(include/det6d_ops.h, section "fused engine ops").  Outputs are caller- or wrapper-allocated
device tensors; every call is asynchronous on the current stream."""
import ctypes
import os

import torch

from .. import _lib as L


#: when set to a list, linear() appends (start_event, end_event) recorded on the launch stream
LINEAR_EVENTS = None
#: when a list: every GEMM-family launch appends (re-issue closure, tensors it keeps alive) — bench.py replays one
#: pass's launches concurrently on many streams to measure the family with the chip full
LINEAR_REPLAY = None


def pack_points(points, ld):
    """(B*N, 1+3+C) [b,x,y,z,f..] -> rows (B*N, ld) [x,y,z,f..,0..]  (pointnet2_backbone.py:193-224)"""
    L.require_cuda(points)
    total, width = points.shape
    ctl = SAMPLER_SEGMENTS
    if ctl is not None and ctl.pack_out is not None:   # a Det6DGroup owns the packed input of its passes
        rows, xyz = ctl.pack_out[0].view(total, ld), ctl.pack_out[1].view(total, 3)
    else:
        rows = torch.empty((total, ld), dtype=torch.float32, device=points.device)
        xyz = torch.empty((total, 3), dtype=torch.float32, device=points.device)
    L.call("det6d_pack_points", total, width - 4, L.ptr(points), ld, L.ptr(rows), L.ptr(xyz), L.stream_ptr())
    return rows, xyz


#: capture controller of runtime.GraphedDet6D (a pass of a Det6DGroup): the SA layers ask it which of their samplers the
#: group launches for all its passes ahead of the captured segments (runtime.hoist_plan) and where their picks live
SAMPLER_SEGMENTS = None


def fps_fused(xyz, lo, hi, m, scores, gamma, idx_out, idx_offset, temp=None, idx_bias=0):
    """one sampler of an SA layer: range slice, sigmoid**gamma weights, 1e10 init, +lo offset and the
    write into the concatenated index buffer all happen inside the kernel"""
    L.require_cuda(xyz, scores, idx_out)
    b, n_total, _ = xyz.shape
    own = temp is None
    if own:
        temp = fps_workspace(b, hi - lo, xyz.device)
    L.call("det6d_fps_fused", b, n_total, lo, hi, m, L.ptr(xyz), L.ptr(scores), float(gamma), L.ptr(temp),
           temp.numel() * temp.element_size(), L.ptr(idx_out), idx_out.shape[1], idx_offset, idx_bias, L.stream_ptr())
    if own and scores is None:
        # a sampler that can fail after its launch (the cooperative one) on a workspace nobody else will look at: its error
        # word goes to the capture controller (a captured pass copies it out after every replay) or, for eager calls, onto
        # the list check_fps_status() drains (Detector3DTemplate.forward calls it where the pass synchronises anyway)
        word = fps_status_word(b, hi - lo, temp)
        if word is not None:
            ctl = SAMPLER_SEGMENTS
            if ctl is not None and hasattr(ctl, 'status_words'):
                ctl.status_words.append(word)
            elif not torch.cuda.is_current_stream_capturing():
                # this launch's word first, then the drain: a drain that raises must not lose the word of the launch that
                # triggered it.  (Nobody drained the list for 64 launches: the check below SYNCHRONISES the current stream —
                # this launch included — and raises FpsTimeout here, for whichever of the 65 launches gave up.)
                PENDING_FPS_STATUS.append(word)
                if len(PENDING_FPS_STATUS) > 64:
                    check_fps_status()


#: error words (int32 device views) of eager cooperative sampler launches that nobody has checked yet
PENDING_FPS_STATUS = []


class FpsTimeout(L.Det6dError):
    """a workgroup of the cooperative 32768 / 65536-point sampler gave up waiting for its partners: the picks of that
    launch are placeholders (include/det6d_ops.h: det6d_fps_fused_status)"""


def check_fps_status(words=None):
    """raises FpsTimeout if a sampler launch behind one of `words` (default: the eager launches since the last call) gave up;
    synchronises the current stream only when there is something to check.  A set word is cleared."""
    pending = PENDING_FPS_STATUS if words is None else words
    if not pending:
        return
    flags = torch.stack([w.reshape(()) for w in pending]).cpu()
    bad = [w for w, f in zip(pending, flags.tolist()) if f]
    if words is None:
        del PENDING_FPS_STATUS[:]
    for w in bad:
        w.zero_()
    if bad:
        raise FpsTimeout("det6d_fps (cooperative): a workgroup waited ~2 s for its partners in %d launch(es); their picks "
                         "are invalid" % len(bad))


def fps_status(b, n, temp):
    """raises if a cooperative sampler launch on `temp` gave up since the last check (synchronises the current stream)"""
    L.call("det6d_fps_fused_status", b, n, L.ptr(temp), temp.numel() * temp.element_size(), L.stream_ptr())


def fps_status_word(b, n, temp):
    """int32 view (1 element) of the sticky error word inside a sampler workspace, or None when the sampler that (b, n)
    selects cannot fail after its launch (det6d_fps_fused_status_offset)"""
    off = int(L.lib().det6d_fps_fused_status_offset(b, n, L.ptr(temp), temp.numel() * temp.element_size()))
    if off < 0:
        return None
    return temp[off:off + 4].view(torch.int32)


def fps_is_cooperative(n):
    """does a d-fps launch over n points per scene use the cooperative multi-workgroup sampler?"""
    return int(L.lib().det6d_fps_fused_workspace_bytes(1, n)) > 4 * n


def fps_workspace(b, n, device='cuda'):
    """scratch of one sampler launch over b scenes of n points (det6d_fps_fused_workspace_bytes: (b, n) floats, more for
    the cooperative sampler of 32768 / 65536-point scenes).  Only the cooperative sampler's sticky error word (the first
    bytes of its workspace: the same place whatever the number of scenes a launch covers) is cleared here — once, when the
    workspace is made; launches never clear it, a status read does."""
    nbytes = int(L.lib().det6d_fps_fused_workspace_bytes(b, n))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=device)
    word = fps_status_word(b, n, ws)
    if word is not None:
        word.zero_()
    return ws


def gather_centres(xyz, idx, rows_out=None, zero_from=0, out=None, idx_bias=0):
    """idx may be a column slice of a wider (b, M) index buffer (row stride M)"""
    L.require_cuda(xyz, rows_out, out)
    if not idx.is_cuda or idx.stride(1) != 1:
        raise L.Det6dError("gather_centres: idx must be a device tensor with unit column stride")
    b, n, _ = xyz.shape
    m = idx.shape[1]
    if out is None:
        out = torch.empty((b, m, 3), dtype=torch.float32, device=xyz.device)
    L.call("det6d_gather_centres", b, n, m, L.ptr(xyz), L.ptr(idx), idx.stride(0) if b > 1 else max(idx.stride(0), m), idx_bias, L.ptr(out),
           L.ptr(rows_out), rows_out.shape[-1] if rows_out is not None else 0, zero_from, L.stream_ptr())
    return out


def with_batch_index(src, ncol=3):
    """(B, M, ld) -> (B*M, 1+ncol) rows [b, src[..., :ncol]]"""
    L.require_cuda(src)
    b, m, ld = src.shape
    dst = torch.empty((b * m, ncol + 1), dtype=torch.float32, device=src.device)
    L.call("det6d_with_batch_index", b, m, L.ptr(src), ld, ncol, L.ptr(dst), L.stream_ptr())
    return dst


def gather_rows(rows_in, idx, ncol, out):
    """out[b,j,:ncol] = rows_in[b, idx[b,j], :ncol]"""
    L.require_cuda(rows_in, idx, out)
    b, n, ld_in = rows_in.shape
    m = idx.shape[1]
    L.call("det6d_gather_rows", b, n, m, ld_in, out.shape[-1], ncol, L.ptr(rows_in), L.ptr(idx), L.ptr(out),
           L.stream_ptr())
    return out


class CompactRows:
    """compact (ragged) row list of one radius group (csrc/compact.hip): only the first
    s = max(smin, 2^ceil(log2 cnt)) slots of every centre are MLP rows; identical pooled features."""
    __slots__ = ('hdr', 'crow_p', 'crow_c', 'capacity', 'ns')

    def __init__(self, hdr, crow_p, crow_c, capacity, ns):
        self.hdr, self.crow_p, self.crow_c, self.capacity, self.ns = hdr, crow_p, crow_c, capacity, ns


#: smallest row class of the compact lists (1, 2 or 4)
COMPACT_SMIN = int(L.experiment_switch('DET6D_COMPACT_SMIN', '1'))
COMPACT_SMALL_CLASSES = True
#: centres with more than g hits take ceil(cnt / g) * g rows cut into power-of-two parts (20 = 16 + 4) whose maxima are
#: combined by an atomic max, instead of being padded to the next power of two; the pooled buffer must be zeroed first
COMPACT_SPLIT = int(L.experiment_switch('DET6D_COMPACT_SPLIT', '1'))   # the granule g (0: padding to the next power of two)


def compact_groups(cnt, idx, n, zero=None):
    """cnt (B,m), idx (B,m,ns) from a ball query over n points per scene -> CompactRows.
    zero = (pooled (B*m, ld), col0, width): that slice of the pooled buffer is cleared by the same call"""
    L.require_cuda(cnt, idx)
    b, m, ns = idx.shape
    cap = int(L.lib().det6d_compact_rows_capacity(b * m, ns))
    hdr = torch.empty((int(L.lib().det6d_compact_hdr_ints(b * m)),), dtype=torch.int32, device=idx.device)
    crow_p = torch.empty((cap,), dtype=torch.int32, device=idx.device)
    crow_c = torch.empty((cap,), dtype=torch.int32, device=idx.device)
    L.call("det6d_compact_groups", b, n, m, ns, min(COMPACT_SMIN, ns), max(COMPACT_SPLIT, min(COMPACT_SMIN, ns)) if COMPACT_SPLIT else 0, L.ptr(cnt), L.ptr(idx),
           L.ptr(hdr), L.ptr(crow_p), L.ptr(crow_c), L.ptr(zero[0]) if zero else None, zero[0].shape[-1] if zero else 0,
           zero[1] if zero else 0, zero[2] if zero else 0, L.stream_ptr())
    return CompactRows(hdr, crow_p, crow_c, cap, ns)


def _alloc_lists(b, m, ns, device):
    cap = int(L.lib().det6d_compact_rows_capacity(b * m, ns))
    hdr = torch.empty((int(L.lib().det6d_compact_hdr_ints(b * m)),), dtype=torch.int32, device=device)
    return CompactRows(hdr, torch.empty((cap,), dtype=torch.int32, device=device),
                       torch.empty((cap,), dtype=torch.int32, device=device), cap, ns)


def compact_groups_pair(found, n, pooled, cols, counted=None):
    """both radius groups of an SA layer: found = [(cnt, idx)] x 2, cols = [(col0, width)] x 2 of `pooled` (cleared).
    counted = the two CompactRows ball_query_pair_lists() has left the per-block part counts in: placement only"""
    (ca, ia), (cb, ib) = found
    L.require_cuda(ca, ia, cb, ib, pooled)
    b, m, _ = ia.shape
    if counted is not None:
        la, lb = counted
    else:
        la, lb = _alloc_lists(b, m, ia.shape[2], ia.device), _alloc_lists(b, m, ib.shape[2], ib.device)
    L.call("det6d_compact_groups_pair_counted" if counted is not None else "det6d_compact_groups_pair",
           b, n, m, COMPACT_SMIN, COMPACT_SPLIT, ia.shape[2], L.ptr(ca), L.ptr(ia), L.ptr(la.hdr),
           L.ptr(la.crow_p), L.ptr(la.crow_c), cols[0][0], cols[0][1], ib.shape[2], L.ptr(cb), L.ptr(ib), L.ptr(lb.hdr),
           L.ptr(lb.crow_p), L.ptr(lb.crow_c), cols[1][0], cols[1][1], L.ptr(pooled), pooled.shape[-1], L.stream_ptr())
    return [la, lb]


def linear(a, w, shift, act, out, k=None, ncols=None, col0=0, idx=None, ctr=None, cnt=None, pool=0, compact=None,
           gather=False, ncols_pad=0):
    """out[..., col0:col0+ncols] = act(A' @ W + shift) with optional neighbour gather / max-pool.

    a:   (R, lda) rows, or (B, n, lda) point rows when `idx` (B, m, ns) is given
    w:   (K_rows, ldw) weights, BN folded;  shift: (ncols,) or None;  act: 0 none / 1 ReLU
    out: (R or R/pool, ldy)
    compact: CompactRows -> the rows are the compact list's (live count read on the device); with `gather` the
         A' rows are gathered through it from the point rows `a` (B, n, lda) and `ctr`; pool = -1 pools by class
    """
    L.require_cuda(a, w, shift, out, idx, ctr, cnt)
    g = L.LinearArgs()
    g.k = w.shape[0] if k is None else k
    g.ncols = w.shape[1] if ncols is None else ncols
    g.a, g.lda = a.data_ptr(), a.shape[-1]
    g.w, g.ldw = w.data_ptr(), w.shape[1]
    g.shift = shift.data_ptr() if shift is not None else None
    g.act = act
    g.y, g.ldy, g.col0 = out.data_ptr(), out.shape[-1], col0
    if compact is not None:
        g.hdr, g.crow_p, g.crow_c = compact.hdr.data_ptr(), compact.crow_p.data_ptr(), compact.crow_c.data_ptr()
        g.rows = compact.capacity
        if gather:
            g.mode = 2
            g.n = a.shape[0] * a.shape[1]
            g.ctr, g.ldctr = ctr.data_ptr(), ctr.shape[-1]
        else:
            g.mode = 0
            assert a.numel() // a.shape[-1] >= compact.capacity
    elif idx is not None:
        bsz, m, ns = idx.shape
        g.mode, g.rows = 1, bsz * m * ns
        g.n, g.m, g.ns = a.shape[1], m, ns
        g.idx = idx.data_ptr()
        g.ctr, g.ldctr = ctr.data_ptr(), ctr.shape[-1]
    else:
        g.mode, g.rows = 0, a.numel() // a.shape[-1]
    g.pool = pool
    g.ncols_pad = ncols_pad      # columns [ncols, ncols_pad) of `out` are zero-filled by the kernel
    g.cnt = cnt.data_ptr() if cnt is not None else None
    if LINEAR_REPLAY is not None:
        def reissue(ptr_of, g=g, a=a, out=out):     # ptr_of: tensor -> device pointer of the replaying stream's own copy
            g2 = L.LinearArgs.from_buffer_copy(g)
            g2.a, g2.y = ptr_of(a), ptr_of(out)
            L.call("det6d_linear", ctypes.byref(g2), L.stream_ptr())
        LINEAR_REPLAY.append((reissue, out, (g, a, w, shift, out, idx, ctr, cnt, compact)))
    if LINEAR_EVENTS is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.call("det6d_linear", ctypes.byref(g), L.stream_ptr())
        e1.record()
        LINEAR_EVENTS.append((e0, e1, g.rows if compact is None else compact.hdr, g.k, g.ncols))
        return out
    L.call("det6d_linear", ctypes.byref(g), L.stream_ptr())
    return out


def group_expand(p, pcol0, w, shift, act, c1, rows_pts, ctr, out, idx=None, compact=None):
    """first layer of a grouped MLP from the per-point partial sums P (csrc/expand.hip): out[r, :c1] for every grouped row
    (dense: idx (B, m, ns); compact: CompactRows), pad columns of `out` zero-filled"""
    L.require_cuda(p, w, shift, rows_pts, ctr, out, idx)
    if compact is not None:
        L.call("det6d_group_expand", compact.capacity, c1, L.ptr(p), p.shape[-1], pcol0, L.ptr(w), w.shape[1], L.ptr(shift), act,
               L.ptr(rows_pts), rows_pts.shape[-1], L.ptr(ctr), ctr.shape[-1], None, 0, 0, 0, L.ptr(compact.hdr),
               L.ptr(compact.crow_p), L.ptr(compact.crow_c), L.ptr(out), out.shape[-1], L.stream_ptr())
    else:
        b, m, ns = idx.shape
        L.call("det6d_group_expand", b * m * ns, c1, L.ptr(p), p.shape[-1], pcol0, L.ptr(w), w.shape[1], L.ptr(shift), act,
               L.ptr(rows_pts), rows_pts.shape[-1], L.ptr(ctr), ctr.shape[-1], L.ptr(idx), rows_pts.shape[1], m, ns, None, None,
               None, L.ptr(out), out.shape[-1], L.stream_ptr())
    return out


#: wide grouped MLPs as one launch (csrc/mlp_group.hip); DET6D_NO_GROUP_KERNEL=1: expand + two GEMM launches
GROUP_KERNEL = L.experiment_switch('DET6D_NO_GROUP_KERNEL') is None


def group_kernel_eligible(layers, ns, compact):
    if not GROUP_KERNEL or len(layers) != 3 or not all(l[3] == 1 for l in layers):
        return False
    return bool(L.lib().det6d_mlp_group3_supported(layers[0][2], layers[1][2], layers[2][2], ns, 1 if compact else 0))


def mlp_group3(p, pcol0, layers, rows_pts, ctr, out, col0, idx=None, cnt=None, compact=None):
    """expand + layer 2 + layer 3 + pool of a wide radius group in one launch"""
    L.require_cuda(p, rows_pts, ctr, out, idx, cnt)
    (w1, s1, c1, _), (w2, s2, c2, _), (w3, s3, c3, _) = layers
    if compact is not None:
        rows, hdr = compact.capacity, compact.hdr
    else:
        b, m, ns = idx.shape
        rows, hdr = b * m * ns, b * m * ns

    def issue(ptr_of=None):
        y = L.ptr(out) if ptr_of is None else ctypes.c_void_p(ptr_of(out))
        if compact is not None:
            # (the kernel draws its tiles from a ticket counter inside the list header: a replay that runs CONCURRENTLY with
            # others of the same launch — bench_legs.family_saturated — needs a header of its own)
            h = L.ptr(compact.hdr) if ptr_of is None else ctypes.c_void_p(ptr_of(compact.hdr))
            tail = (None, 0, 0, 0, None, h, L.ptr(compact.crow_p), L.ptr(compact.crow_c))
        else:
            tail = (L.ptr(idx), rows_pts.shape[1], m, ns, L.ptr(cnt), None, None, None)
        L.call("det6d_mlp_group3", rows, L.ptr(p), p.shape[-1], pcol0, L.ptr(w1), w1.shape[1], L.ptr(s1), c1, L.ptr(w2), w2.shape[1],
               L.ptr(s2), c2, L.ptr(w3), w3.shape[1], L.ptr(s3), c3, L.ptr(rows_pts), rows_pts.shape[-1], L.ptr(ctr), ctr.shape[-1],
               *tail, y, out.shape[-1], col0, L.stream_ptr())
    ev = None
    if LINEAR_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if LINEAR_REPLAY is not None:
        LINEAR_REPLAY.append((issue, out, (p, layers, rows_pts, ctr, out, idx, cnt, compact), [compact.hdr] if compact is not None else []))
    issue()
    if ev is not None:
        ev[1].record()
        LINEAR_EVENTS.append((ev[0], ev[1], hdr, 1, c1 * c2 + c2 * c3))
    return out


#: short stacks of plain layers in one launch (csrc/mlp_rows.hip); DET6D_NO_ROWS_KERNEL=1: one det6d_linear per layer
ROWS_KERNEL = L.experiment_switch('DET6D_NO_ROWS_KERNEL') is None


def _rows_descriptors(chains, ptr_of=None):
    flat = [l for chain in chains for l in chain]
    arr = (L.RowsLayer * len(flat))()
    for d, (w, wrow0, shift, k, n, act, out, ocol0) in zip(arr, flat):
        d.w, d.ldw, d.wrow0 = w.data_ptr(), w.shape[1], wrow0
        d.shift = shift.data_ptr() if shift is not None else None
        d.k, d.n, d.act = k, n, act
        if out is not None:
            d.out, d.ldo, d.ocol0 = (out.data_ptr() if ptr_of is None else ptr_of(out)), out.shape[-1], ocol0
    return flat, (ctypes.c_int * len(chains))(*[len(c) for c in chains]), arr


def mlp_rows_eligible(k0, chains):
    """chains: [[(w, wrow0, shift, k, n, act, out, ocol0), ...], ...]; the library decides (det6d_mlp_rows_supported: chain
    structure, widths and the LDS a tile's activation buffers take), so a stack that does not fit goes through run_chain's
    one-launch-per-layer route instead of failing inside the forward pass or a graph capture"""
    if not ROWS_KERNEL or not 1 <= len(chains) <= 2 or not all(1 <= len(c) <= 4 and c[0][3] == k0 for c in chains):
        return False
    if any(l[5] not in (0, 1) for c in chains for l in c):
        return False
    _, counts, arr = _rows_descriptors(chains)
    return bool(L.lib().det6d_mlp_rows_supported(len(chains), counts, arr))


def mlp_rows(x, xcol0, chains):
    """x (R, ldx) device rows; every chain reads columns [xcol0, xcol0 + k0) of it; see mlp_rows_eligible for the spec"""
    L.require_cuda(x)
    rows = x.numel() // x.shape[-1]
    flat = [l for chain in chains for l in chain]

    def issue(ptr_of=None):
        _, counts, arr = _rows_descriptors(chains, ptr_of)
        L.call("det6d_mlp_rows", rows, L.ptr(x) if ptr_of is None else ctypes.c_void_p(ptr_of(x)), x.shape[-1], xcol0, len(chains),
               counts, arr, L.stream_ptr())
    ev = None
    if LINEAR_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if LINEAR_REPLAY is not None:
        outs = [l[6] for l in flat if l[6] is not None]
        LINEAR_REPLAY.append((issue, outs[-1], (x, chains)))
    issue()
    if ev is not None:
        ev[1].record()
        LINEAR_EVENTS.append((ev[0], ev[1], rows, 1, sum(l[3] * l[4] for l in flat)))


def sigmoid_pow(scores, gamma, out=None):
    L.require_cuda(scores)
    out = torch.empty_like(scores) if out is None else out
    L.call("det6d_sigmoid_pow", scores.numel(), L.ptr(scores), float(gamma), L.ptr(out), L.stream_ptr())
    return out


def vote_points(off, cand, rng, vote, off_out=None):
    L.require_cuda(off, cand, vote, off_out)
    rows = off.numel() // off.shape[-1]
    L.call("det6d_vote_points", rows, L.ptr(off), off.shape[-1], L.ptr(cand), cand.shape[-1],
           float(rng[0]), float(rng[1]), float(rng[2]), L.ptr(vote), vote.shape[-1], L.ptr(off_out),
           L.stream_ptr())
    return vote


def decode_boxes(code, pts, nbin, ground_aware, minus, threshold_rad, factor_rad, out=None):
    L.require_cuda(code, pts)
    rows = code.numel() // code.shape[-1]
    out = torch.empty((rows, 9), dtype=torch.float32, device=code.device) if out is None else out
    L.call("det6d_decode_boxes", rows, nbin, int(ground_aware), int(minus), float(threshold_rad),
           float(factor_rad), L.ptr(code), code.shape[-1], L.ptr(pts), pts.shape[-1], L.ptr(out), L.stream_ptr())
    return out


def postprocess(cls, boxes, b, score_thr, pre_max, post_max, nms_thr):
    L.require_cuda(cls, boxes)
    p = cls.shape[0] // b
    dev = cls.device
    ob = torch.empty((b, post_max, 9), dtype=torch.float32, device=dev)
    os_ = torch.empty((b, post_max), dtype=torch.float32, device=dev)
    ol = torch.empty((b, post_max), dtype=torch.int32, device=dev)
    oi = torch.empty((b, post_max), dtype=torch.int32, device=dev)
    oc = torch.empty((b,), dtype=torch.int32, device=dev)
    ws = torch.empty((int(L.lib().det6d_postprocess_workspace_bytes(b)),), dtype=torch.uint8, device=dev)
    L.call("det6d_postprocess", b, p, cls.shape[1], L.ptr(cls), L.ptr(boxes), float(score_thr), pre_max,
           post_max, float(nms_thr), L.ptr(ws), L.ptr(ob), L.ptr(os_), L.ptr(ol), L.ptr(oi), L.ptr(oc), L.stream_ptr())
    return ob, os_, ol, oi, oc


def group_maxpool(x, ns, ncols, cnt, out, col0):
    """out[r, col0:col0+ncols] = (cnt[r] > 0) * max over the ns rows of group r of x[:, :ncols] — mask + max_pool2d for any
    nsample (pointnet2_modules.py:465-472); the GEMM epilogues pool nsample in {8, 16, 32} themselves"""
    L.require_cuda(x, cnt, out)
    groups = x.shape[0] // ns
    L.call("det6d_group_maxpool", groups, ns, ncols, L.ptr(x), x.shape[-1], L.ptr(cnt), L.ptr(out), out.shape[-1], col0,
           L.stream_ptr())


def nms_device(boxes, thresh, normal=False):
    """device-resident NMS: returns (keep int64 (K,), num_keep int32 (1,)) without a host sync"""
    L.require_cuda(boxes)
    k = boxes.shape[0]
    words = max(int(L.lib().det6d_nms_mask_words(k)), 1)
    mask = torch.empty((words,), dtype=torch.int64, device=boxes.device)
    keep = torch.empty((max(k, 1),), dtype=torch.int64, device=boxes.device)
    num = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    L.call("det6d_nms_normal" if normal else "det6d_nms", k, L.ptr(boxes), float(thresh), L.ptr(mask),
           L.ptr(keep), L.ptr(num), L.stream_ptr())
    return keep, num


#: point count from which the grid-hashed search replaces the brute-force sweep
GRID_QUERY_MIN_N = int(L.experiment_switch('DET6D_GRID_MIN_N', '2048'))


def ball_query_pair(xyz, new_xyz, shell_a, shell_b, grid=None):
    """both radius groups of an SA layer in one launch; shell = (radius_in, radius_out, nsample).
    Returns (cnt_a, idx_a, cnt_b, idx_b), all int32, fully written by the kernel.  Large point sets go
    through the grid-hashed kernel (identical results), small ones through the brute-force sweep."""
    L.require_cuda(xyz, new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    dev = xyz.device
    if grid is None:
        grid = GRID_QUERY_MIN_N <= n
    if grid and L.lib().det6d_ball_query_grid_supported(n, shell_a[2], shell_b[2]):   # (it ranks one list entry per lane: nsample <= 64)
        cnt_a = torch.empty((b, m), dtype=torch.int32, device=dev)
        cnt_b = torch.empty((b, m), dtype=torch.int32, device=dev)
        idx_a = torch.empty((b, m, shell_a[2]), dtype=torch.int32, device=dev)
        idx_b = torch.empty((b, m, shell_b[2]), dtype=torch.int32, device=dev)
        ws = torch.empty((int(L.lib().det6d_ball_query_grid_workspace_bytes(b, n)),), dtype=torch.uint8, device=dev)
        L.call("det6d_ball_query_pair_grid", b, n, m, float(shell_a[0]), float(shell_a[1]), shell_a[2],
               float(shell_b[0]), float(shell_b[1]), shell_b[2], L.ptr(new_xyz), L.ptr(xyz), L.ptr(ws), L.ptr(cnt_a),
               L.ptr(idx_a), L.ptr(cnt_b), L.ptr(idx_b), L.stream_ptr())
        return cnt_a, idx_a, cnt_b, idx_b
    if max(shell_a[2], shell_b[2]) > 128:              # beyond the fused kernel's static hit lists: one query per shell
        out = []
        for rin, rout, ns in (shell_a, shell_b):
            cnt = torch.zeros((b, m), dtype=torch.int32, device=dev)
            idx = torch.zeros((b, m, ns), dtype=torch.int32, device=dev)
            L.call("det6d_ball_query_dilated", b, n, m, float(rin), float(rout), ns, L.ptr(new_xyz), L.ptr(xyz), L.ptr(cnt),
                   L.ptr(idx), L.stream_ptr())
            out += [cnt, idx]
        return tuple(out)
    cnt_a = torch.empty((b, m), dtype=torch.int32, device=dev)
    cnt_b = torch.empty((b, m), dtype=torch.int32, device=dev)
    idx_a = torch.empty((b, m, shell_a[2]), dtype=torch.int32, device=dev)
    idx_b = torch.empty((b, m, shell_b[2]), dtype=torch.int32, device=dev)
    L.call("det6d_ball_query_pair", b, n, m, float(shell_a[0]), float(shell_a[1]), shell_a[2], float(shell_b[0]),
           float(shell_b[1]), shell_b[2], L.ptr(new_xyz), L.ptr(xyz), L.ptr(cnt_a), L.ptr(idx_a), L.ptr(cnt_b),
           L.ptr(idx_b), L.stream_ptr())
    return cnt_a, idx_a, cnt_b, idx_b


def ball_query_pair_lists(xyz, new_xyz, shell_a, shell_b):
    """The grid query as the compact-row engine uses it (csrc/ball_query_grid.hip: det6d_ball_query_pair_grid_lists): same
    hits and counts as ball_query_pair(); index rows written only as far as the list builder reads them (no padding beyond
    the next power of two >= max(cnt, 4)), and the builder's per-block part counts left in the two CompactRows it returns
    — compact_groups_pair(..., counted=them) then only places.  Returns None when the shape does not qualify (small clouds,
    m not a multiple of 256, nsample not 4 / 8 / 16 / 32): callers use ball_query_pair() + compact_groups_pair()."""
    L.require_cuda(xyz, new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    dev = xyz.device
    ns_a, ns_b = shell_a[2], shell_b[2]
    if (n < GRID_QUERY_MIN_N or m % 256 or not COMPACT_SPLIT or COMPACT_SMIN > 4 or not all(ns in (4, 8, 16, 32) for ns in (ns_a, ns_b))
            or not L.lib().det6d_ball_query_grid_supported(n, ns_a, ns_b)):
        return None
    cnt_a = torch.empty((b, m), dtype=torch.int32, device=dev)
    cnt_b = torch.empty((b, m), dtype=torch.int32, device=dev)
    idx_a = torch.empty((b, m, ns_a), dtype=torch.int32, device=dev)
    idx_b = torch.empty((b, m, ns_b), dtype=torch.int32, device=dev)
    ws = torch.empty((int(L.lib().det6d_ball_query_grid_workspace_bytes(b, n)),), dtype=torch.uint8, device=dev)
    la, lb = _alloc_lists(b, m, ns_a, dev), _alloc_lists(b, m, ns_b, dev)
    L.call("det6d_ball_query_pair_grid_lists", b, n, m, float(shell_a[0]), float(shell_a[1]), ns_a, float(shell_b[0]), float(shell_b[1]),
           ns_b, L.ptr(new_xyz), L.ptr(xyz), L.ptr(ws), L.ptr(cnt_a), L.ptr(idx_a), L.ptr(cnt_b), L.ptr(idx_b), COMPACT_SMIN,
           COMPACT_SPLIT, L.ptr(la.hdr), L.ptr(lb.hdr), L.stream_ptr())
    return cnt_a, idx_a, cnt_b, idx_b, la, lb


#: route [68 -> 64 -> 64|96 -> 128] groups through the wide register chain kernel
CHAIN_WIDE = L.experiment_switch('DET6D_CHAIN_NO_WIDE') is None   # the C entry honours the same switch


def chain_eligible(lda, layers, ns):
    """can a grouped 3-layer MLP run as ONE det6d_mlp_chain3 launch?  layers: [(W, shift, cout, act)] x 3"""
    if len(layers) != 3 or ns not in (16, 32) or not all(l[3] == 1 for l in layers):
        return False
    c1, c2, c3 = layers[0][2], layers[1][2], layers[2][2]
    if lda == 68 and c1 == 64 and c2 in (64, 96) and c3 == 128:      # wide register chain (SA2-sized groups)
        return CHAIN_WIDE
    return lda <= 8 and c1 <= 32 and c2 <= 32 and c3 <= 64


#: compact-row groups through the register chain kernels (DET6D_COMPACT_NO_CHAIN=1: three det6d_linear launches)
COMPACT_CHAIN = L.experiment_switch('DET6D_COMPACT_NO_CHAIN') is None


def chain_compact_eligible(lda, layers):
    if not COMPACT_CHAIN or len(layers) != 3 or not all(l[3] == 1 for l in layers):
        return False
    c = (layers[0][2], layers[1][2], layers[2][2])
    if lda == 68 and c[0] == 64 and c[1] in (64, 96) and c[2] == 128:
        return CHAIN_WIDE
    return lda == 4 and c in ((16, 16, 32), (32, 32, 64))


def mlp_chain3_compact(rows_pts, cr, ctr, layers, out, col0):
    """mlp_chain3 over a CompactRows list"""
    L.require_cuda(rows_pts, ctr, out)
    (w1, s1, c1, _), (w2, s2, c2, _), (w3, s3, c3, _) = layers
    ev = None
    if LINEAR_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    def issue(ptr_of=None):
        y = L.ptr(out) if ptr_of is None else ctypes.c_void_p(ptr_of(out))
        L.call("det6d_mlp_chain3_compact", cr.capacity, L.ptr(cr.hdr), L.ptr(cr.crow_p), L.ptr(cr.crow_c), L.ptr(rows_pts),
               rows_pts.shape[-1], L.ptr(ctr), ctr.shape[-1], L.ptr(w1), w1.shape[1], L.ptr(s1), c1, L.ptr(w2), w2.shape[1],
               L.ptr(s2), c2, L.ptr(w3), w3.shape[1], L.ptr(s3), c3, y, out.shape[-1], col0, L.stream_ptr())
    if LINEAR_REPLAY is not None:
        LINEAR_REPLAY.append((issue, out, (rows_pts, cr, ctr, layers, out)))
    issue()
    if ev is not None:
        ev[1].record()
        LINEAR_EVENTS.append((ev[0], ev[1], cr.hdr, 1, (rows_pts.shape[-1] * c1 + c1 * c2 + c2 * c3)))
    return out


def mlp_chain3(rows_pts, idx, ctr, cnt, layers, out, col0):
    """gather + 3 x (GEMM, shift, ReLU) + mask + max-pool in one launch (narrow widths only)"""
    L.require_cuda(rows_pts, idx, ctr, cnt, out)
    b, m, ns = idx.shape
    (w1, s1, c1, _), (w2, s2, c2, _), (w3, s3, c3, _) = layers
    ev = None
    if LINEAR_EVENTS is not None:   # the chain is part of the MLP GEMM family bench.py prices
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    L.call("det6d_mlp_chain3", b * m * ns, rows_pts.shape[1], m, ns, L.ptr(rows_pts), rows_pts.shape[-1], L.ptr(idx),
           L.ptr(ctr), ctr.shape[-1], L.ptr(cnt), L.ptr(w1), w1.shape[1], L.ptr(s1), c1, L.ptr(w2), w2.shape[1],
           L.ptr(s2), c2, L.ptr(w3), w3.shape[1], L.ptr(s3), c3, L.ptr(out), out.shape[-1], col0, L.stream_ptr())
    if ev is not None:
        ev[1].record()
        r = b * m * ns
        LINEAR_EVENTS.append((ev[0], ev[1], r, 1, (rows_pts.shape[-1] * c1 + c1 * c2 + c2 * c3)))
    return out


def prepare_points(raw, offsets, point_cloud_range, num_points, seed, scene_ids=None, near_depth=40.0, out=None,
                   workspace=None, n_in=None):
    """raw (total_raw, C) concatenated frames + offsets (B+1) int32 -> (points (B*N, 1+C), n_in_range (B))
    range mask + sample_points + collate batch index in one launch (include/det6d_ops.h: input producer).
    `out` / `workspace` / `n_in` may be passed to reuse buffers (e.g. a captured graph's static input)."""
    L.require_cuda(raw, offsets, scene_ids, out, workspace, n_in)
    total_raw, c = raw.shape
    b = offsets.numel() - 1
    dev = raw.device
    need = int(L.lib().det6d_prepare_points_workspace_bytes(b, total_raw))
    ws = workspace if workspace is not None and workspace.numel() >= need else torch.empty((need,), dtype=torch.uint8, device=dev)
    if out is None:
        out = torch.empty((b * num_points, 1 + c), dtype=torch.float32, device=dev)
    if n_in is None:
        n_in = torch.empty((b,), dtype=torch.int32, device=dev)
    r = [float(v) for v in point_cloud_range]
    L.call("det6d_prepare_points", b, L.ptr(offsets), L.ptr(scene_ids), total_raw, c, L.ptr(raw), r[0], r[1], r[3], r[4],
           int(num_points), float(near_depth), ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), L.ptr(ws), L.ptr(out),
           L.ptr(n_in), L.stream_ptr())
    return out, n_in


def kitti_annos(boxes, scene_of, calib):
    """boxes (T, >=7) LiDAR detections, scene_of (T) int32, calib (B, 28) -> (T, 12)
    [camera box x,y,z,l,h,w,ry | image box x1,y1,x2,y2 | alpha]  (include/det6d_ops.h: output consumer)"""
    L.require_cuda(boxes, scene_of, calib)
    total, ld = boxes.shape
    out = torch.empty((total, 12), dtype=torch.float32, device=boxes.device)
    L.call("det6d_kitti_annos", total, L.ptr(boxes), ld, L.ptr(scene_of), L.ptr(calib), L.ptr(out), L.stream_ptr())
    return out


def make_slope(points, boxes9, params):
    """in-place SlopeAug geometry: points (N, >=3) float32 cuda, boxes9 (M, 9) float64 cuda (or None),
    params 16 host doubles (include/det6d_ops.h: SlopeAug geometry)"""
    import numpy as np
    L.require_cuda(points, boxes9)
    prm = np.ascontiguousarray(params, np.float64)
    assert prm.size == 16 and points.dtype == torch.float32 and (boxes9 is None or boxes9.dtype == torch.float64)
    n_boxes = 0 if boxes9 is None else boxes9.shape[0]
    L.call("det6d_make_slope", points.shape[0], L.ptr(points), points.shape[1], n_boxes, L.ptr(boxes9),
           prm.ctypes.data_as(ctypes.c_void_p), L.stream_ptr())


def boxes9_corners(boxes9):
    """(M, 9) float64 cuda [x,y,z,dx,dy,dz,rz,ry,rx] -> (M, 8, 3) float64 corners"""
    L.require_cuda(boxes9)
    assert boxes9.dtype == torch.float64
    out = torch.empty((boxes9.shape[0], 8, 3), dtype=torch.float64, device=boxes9.device)
    L.call("det6d_boxes9_corners", boxes9.shape[0], L.ptr(boxes9), L.ptr(out), L.stream_ptr())
    return out
