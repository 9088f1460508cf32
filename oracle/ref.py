"""ctypes front-end of oracle/_ref/libref_iou3d_cpu.so — the REFERENCE's own
core/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp compiled where it lies (oracle/Makefile target _ref).
Test infrastructure only; available() is False when the prebuilt library did not travel."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_iou3d_cpu.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (libtorch must be resident before the dlopen)
        _lib = ctypes.CDLL(_PATH)
    return _lib


def boxes_iou_bev_cpu(boxes_a, boxes_b):
    a = np.ascontiguousarray(boxes_a, np.float32)
    b = np.ascontiguousarray(boxes_b, np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    lib().ref_boxes_iou_bev_cpu(a.ctypes.data_as(fp), a.shape[0], b.ctypes.data_as(fp), b.shape[0],
                                out.ctypes.data_as(fp))
    return out
