"""de6d_amd — MI355X-native (gfx950) Det6D inference hot path.

Layout:
  csrc/      hand-written HIP kernels + the C ABI (include/det6d_ops.h) -> libdet6d_hip.so
  _lib.py    ctypes binding (torch tensors only lend device pointers and the stream)
  ops/       drop-in replacements for the reference's two extension modules
             (pointnet2_batch_cuda, iou3d_nms_cuda) and its Python op wrappers
  pcdet/     host-side mirror of the reference's config / registry / model API for this path
"""
__version__ = "0.1.0"
