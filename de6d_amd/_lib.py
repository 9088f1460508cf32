"""ctypes binding of csrc/libdet6d_hip.so (C ABI: include/det6d_ops.h).

PyTorch is plumbing here: tensors provide device memory (``data_ptr()``) and the current HIP
stream; nothing below ever computes with torch ops.  There is NO fallback: if the library is
missing or a call fails, an exception is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdet6d_hip.so")
if os.environ.get("DET6D_KNOBS_LIB"):            # scripts only: the shipped kernels with the route switches live (-DDET6D_KNOBS)
    LIB_PATH = os.path.join(_HERE, "csrc", "libdet6d_hip_knobs.so")
    if os.environ["DET6D_KNOBS_LIB"].endswith(".so"):      # an A/B against another BUILD of the library (scripts/r06/gpu_t14.sh)
        LIB_PATH = os.environ["DET6D_KNOBS_LIB"]
elif os.environ.get("DET6D_EXPERIMENTS_LIB"):      # scripts/experiments only: the -DDET6D_EXPERIMENTS build (de6d_amd/_build.py)
    LIB_PATH = os.path.join(_HERE, "csrc", "libdet6d_hip_experiments.so")

c_int, c_float, c_void_p, c_int64 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int64


class Det6dError(RuntimeError):
    pass


class LinearArgs(ctypes.Structure):
    """mirror of det6d_linear_args (include/det6d_ops.h)"""
    _fields_ = [("mode", c_int), ("rows", c_int), ("k", c_int), ("ncols", c_int),
                ("a", c_void_p), ("lda", c_int),
                ("w", c_void_p), ("ldw", c_int),
                ("shift", c_void_p),
                ("act", c_int),
                ("y", c_void_p), ("ldy", c_int), ("col0", c_int),
                ("n", c_int), ("m", c_int), ("ns", c_int),
                ("idx", c_void_p),
                ("ctr", c_void_p), ("ldctr", c_int),
                ("pool", c_int),
                ("cnt", c_void_p),
                ("hdr", c_void_p), ("crow_p", c_void_p), ("crow_c", c_void_p), ("ncols_pad", c_int)]


_P = c_void_p
class RowsLayer(ctypes.Structure):
    """det6d_rows_layer (include/det6d_ops.h)"""
    _fields_ = [("w", c_void_p), ("ldw", c_int), ("wrow0", c_int), ("shift", c_void_p), ("k", c_int), ("n", c_int), ("act", c_int),
                ("out", c_void_p), ("ldo", c_int), ("ocol0", c_int)]


class EvalMatchArgs(ctypes.Structure):
    """det6d_eval_match_args (include/det6d_ops.h)"""
    _fields_ = [("n_frames", ctypes.c_int), ("n_thresh", ctypes.c_int), ("metric", ctypes.c_int),
                ("compute_aos", ctypes.c_int), ("dt_f32", ctypes.c_int), ("min_overlap", ctypes.c_double),
                ("thresholds", ctypes.c_void_p), ("dt_off", ctypes.c_void_p), ("gt_off", ctypes.c_void_p),
                ("dc_off", ctypes.c_void_p), ("pair_off", ctypes.c_void_p), ("overlaps", ctypes.c_void_p),
                ("gt_alpha", ctypes.c_void_p), ("dt_bbox", ctypes.c_void_p), ("dt_alpha", ctypes.c_void_p),
                ("dt_score", ctypes.c_void_p), ("ignored_gt", ctypes.c_void_p), ("ignored_dt", ctypes.c_void_p),
                ("dc_bbox", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("stats", ctypes.c_void_p),
                ("tp_scores", ctypes.c_void_p), ("tp_count", ctypes.c_void_p), ("gt_of_tp", ctypes.c_void_p)]


_SIGNATURES = {
    "det6d_fps": [c_int, c_int, c_int, _P, _P, _P, _P],
    "det6d_fps_weights": [c_int, c_int, c_int, _P, _P, _P, _P, _P],
    "det6d_gather_points": [c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "det6d_gather_points_grad": [c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "det6d_ball_query": [c_int, c_int, c_int, c_float, c_int, _P, _P, _P, _P],
    "det6d_ball_query_cnt": [c_int, c_int, c_int, c_float, c_int, _P, _P, _P, _P, _P],
    "det6d_ball_query_dilated": [c_int, c_int, c_int, c_float, c_float, c_int, _P, _P, _P, _P, _P],
    "det6d_ball_query_pair": [c_int, c_int, c_int, c_float, c_float, c_int, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P, _P],
    "det6d_ball_query_pair_grid": [c_int, c_int, c_int, c_float, c_float, c_int, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P, _P, _P],
    "det6d_ball_query_pair_grid_lists": [c_int, c_int, c_int, c_float, c_float, c_int, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P, _P,
                                         c_int, c_int, _P, _P, _P],
    "det6d_group_points": [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "det6d_group_points_grad": [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "det6d_three_nn": [c_int, c_int, c_int, _P, _P, _P, _P, _P],
    "det6d_three_interpolate": [c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P],
    "det6d_three_interpolate_grad": [c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P],
    "det6d_boxes_overlap_bev": [c_int, _P, c_int, _P, _P, _P],
    "det6d_boxes_iou_bev": [c_int, _P, c_int, _P, _P, _P],
    "det6d_boxes_iou_bev_cpu": [c_int, _P, c_int, _P, _P],
    "det6d_nms": [c_int, _P, c_float, _P, _P, _P, _P],
    "det6d_nms_normal": [c_int, _P, c_float, _P, _P, _P, _P],
    "det6d_nms_to_host": [c_int, _P, c_float, _P, c_int, _P],
    "det6d_pack_points": [c_int, c_int, _P, c_int, _P, _P, _P],
    "det6d_fps_fused_status": [c_int, c_int, _P, c_int64, _P],
    "det6d_fps_fused": [c_int, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, c_int64, _P, c_int, c_int, c_int, _P],
    "det6d_gather_centres": [c_int, c_int, c_int, _P, _P, c_int, c_int, _P, _P, c_int, c_int, _P],
    "det6d_with_batch_index": [c_int, c_int, _P, c_int, c_int, _P, _P],
    "det6d_gather_rows": [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "det6d_linear": [ctypes.POINTER(LinearArgs), _P],
    "det6d_mlp_chain3": [c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, _P, c_int, _P, c_int, _P, c_int, _P, c_int,
                         _P, c_int, _P, c_int, _P, c_int, c_int, _P],
    "det6d_mlp_chain3_compact": [c_int, _P, _P, _P, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int,
                                 _P, c_int, _P, c_int, c_int, _P],
    "det6d_compact_groups_pair": [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P,
                                  _P, c_int, c_int, _P, c_int, _P],
    "det6d_compact_groups_pair_counted": [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P,
                                          _P, c_int, c_int, _P, c_int, _P],
    "det6d_compact_groups": [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "det6d_mlp_group3_supported": [c_int, c_int, c_int, c_int, c_int],
    "det6d_mlp_group3": [c_int, _P, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int,
                         _P, c_int, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P],
    "det6d_mlp_rows": [c_int, _P, c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(RowsLayer), _P],
    "det6d_group_expand": [c_int, c_int, _P, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int,
                           _P, _P, _P, _P, c_int, _P],
    "det6d_sigmoid_pow": [c_int, _P, c_float, _P, _P],
    "det6d_group_maxpool": [c_int, c_int, c_int, _P, c_int, _P, _P, c_int, c_int, _P],
    "det6d_vote_points": [c_int, _P, c_int, _P, c_int, c_float, c_float, c_float, _P, c_int, _P, _P],
    "det6d_decode_boxes": [c_int, c_int, c_int, c_int, c_float, c_float, _P, c_int, _P, c_int, _P, _P],
    "det6d_prepare_points": [c_int, _P, _P, c_int, c_int, _P, c_float, c_float, c_float, c_float, c_int, c_float,
                             ctypes.c_uint64, _P, _P, _P, _P],
    "det6d_kitti_annos": [c_int, _P, c_int, _P, _P, _P, _P],
    "det6d_make_slope": [c_int, _P, c_int, c_int, _P, _P, _P],
    "det6d_boxes9_corners": [c_int, _P, _P, _P],
    "det6d_eval_overlaps": [c_int, c_int, _P, _P, _P, c_int64, _P, _P, c_int, _P, _P],
    "det6d_eval_match": [ctypes.POINTER(EvalMatchArgs), _P],
    "det6d_eval_reduce": [c_int, c_int, _P, _P, _P],
    "det6d_postprocess": [c_int, c_int, c_int, _P, _P, c_float, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P, _P],
}

#: every symbol include/det6d_ops.h declares (tests/test_boundary.py checks the export table)
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["det6d_version", "det6d_last_error", "det6d_nms_mask_words",
                                                  "det6d_postprocess_workspace_bytes", "det6d_fps_fused_workspace_bytes",
                                                  "det6d_fps_fused_status_offset", "det6d_mlp_rows_supported",
                                                  "det6d_ball_query_grid_supported",
                                                  "det6d_ball_query_grid_workspace_bytes",
                                                  "det6d_prepare_points_workspace_bytes",
                                                  "det6d_compact_rows_capacity", "det6d_compact_hdr_ints"])

_lib = None


def experiment_switch(name, default=None):
    """value of an alternative-route variable (DET6D_NO_EXPAND, DET6D_COMPACT_SPLIT, ...) — honoured only together with
    DET6D_EXPERIMENTS_LIB=1, i.e. in the experiments build; the shipped configuration has ONE route per shape"""
    if not (os.environ.get("DET6D_EXPERIMENTS_LIB") or os.environ.get("DET6D_KNOBS_LIB")):
        return default
    return os.environ.get(name, default)


def lib():
    """Load libdet6d_hip.so (after torch, so both share torch's libamdhip64.so.7)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Det6dError(
                "libdet6d_hip.so is missing (%s). Build it with `python -m de6d_amd._build` "
                "(or __graft_entry__.build()); there is no CPU/PyTorch fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = c_int
        handle.det6d_version.restype = ctypes.c_char_p
        handle.det6d_last_error.restype = ctypes.c_char_p
        handle.det6d_nms_mask_words.argtypes = [c_int]
        handle.det6d_nms_mask_words.restype = c_int64
        handle.det6d_ball_query_grid_workspace_bytes.argtypes = [c_int, c_int]
        handle.det6d_ball_query_grid_workspace_bytes.restype = c_int64
        handle.det6d_prepare_points_workspace_bytes.argtypes = [c_int, c_int]
        handle.det6d_prepare_points_workspace_bytes.restype = c_int64
        handle.det6d_fps_fused_workspace_bytes.argtypes = [c_int, c_int]
        handle.det6d_fps_fused_workspace_bytes.restype = c_int64
        handle.det6d_mlp_rows_supported.argtypes = [c_int, _P, _P]
        handle.det6d_mlp_rows_supported.restype = c_int
        handle.det6d_fps_fused_status_offset.argtypes = [c_int, c_int, _P, c_int64]
        handle.det6d_fps_fused_status_offset.restype = c_int64
        handle.det6d_ball_query_grid_supported.argtypes = [c_int, c_int, c_int]
        handle.det6d_ball_query_grid_supported.restype = c_int
        handle.det6d_compact_hdr_ints.argtypes = [c_int]
        handle.det6d_compact_hdr_ints.restype = c_int
        handle.det6d_compact_rows_capacity.argtypes = [c_int, c_int]
        handle.det6d_compact_rows_capacity.restype = c_int
        handle.det6d_postprocess_workspace_bytes.argtypes = [c_int]
        handle.det6d_postprocess_workspace_bytes.restype = c_int64
        _lib = handle
    return _lib


def version():
    return lib().det6d_version().decode()


def stream_ptr(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def check(rc, what):
    if rc < 0:
        raise Det6dError("%s failed: rc=%d %s" % (what, rc, lib().det6d_last_error().decode()))
    return rc


def call(name, *args):
    """Invoke a C-ABI entry point on the current stream; raises on any negative status."""
    return check(getattr(lib(), name)(*args), name)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise Det6dError("det6d ops need device tensors (got a %s tensor); there is no CPU path"
                             % t.device)
        if t is not None and not t.is_contiguous():
            raise Det6dError("det6d ops need contiguous tensors")
