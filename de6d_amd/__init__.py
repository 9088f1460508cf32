"""de6d_amd — MI355X-native (gfx950) Det6D inference hot path.

Layout:
  csrc/      hand-written HIP kernels + the C ABI (include/det6d_ops.h) -> libdet6d_hip.so
  _lib.py    ctypes binding (torch tensors only lend device pointers and the stream)
  ops/       drop-in replacements for the reference's two extension modules
             (pointnet2_batch_cuda, iou3d_nms_cuda) and its Python op wrappers
  pcdet/     host-side mirror of the reference's config / registry / model API for this path
"""
__version__ = "0.1.0"

import os as _os
import sys as _sys


def _export_hw_queues():
    """ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable when the runtime
    initialises.  The throughput pipeline (runtime.ScenePipeline) keeps 16 + 6 streams busy; on 4 queues it runs 4x slower.
    So: when this process has not initialised the GPU yet and the caller has not chosen a value, export 24 here; when it HAS
    (the ROS-node shape: sim/gazebo/src/detection/script/detection.py:108-126 builds its CUDA context first), remember what
    was in effect — runtime.require_hw_queues() raises instead of running aliased."""
    torch = _sys.modules.get('torch')
    if torch is not None and torch.cuda.is_initialized():
        return int(_os.environ.get('GPU_MAX_HW_QUEUES', '4'))
    _os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
    return None


#: hardware queues in effect if HIP was already initialised when the package was imported, else None (the variable is live)
HW_QUEUES_AT_IMPORT = _export_hw_queues()


#: where the reference imports its two compiled extension modules from
#: (core/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py:7, core/pcdet/ops/iou3d_nms/iou3d_nms_utils.py:9)
PCDET_EXTENSION_SITES = {
    'pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda': 'de6d_amd.ops.pointnet2_batch_hip',
    'pcdet.ops.iou3d_nms.iou3d_nms_cuda': 'de6d_amd.ops.iou3d_nms_hip',
}


def install_pcdet_ops():
    """Zero-edit drop-in: registers the two HIP-backed modules in ``sys.modules`` under the names the REFERENCE imports its
    compiled extensions by, so that an unmodified reference checkout runs on libdet6d_hip.so:

        import de6d_amd; de6d_amd.install_pcdet_ops()
        from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils      # `from . import pointnet2_batch_cuda` finds ours
        from pcdet.ops.iou3d_nms import iou3d_nms_utils                       # `from . import iou3d_nms_cuda` likewise

    Call it before the first import of those reference modules.  Same names, positional arities and return conventions as
    the pybind modules (tests/golden/extension_api.json, tests/test_boundary.py); kernels run on torch's current stream.
    Returns the two modules."""
    import importlib
    import sys
    mods = []
    for site, ours in PCDET_EXTENSION_SITES.items():
        mod = importlib.import_module(ours)
        sys.modules[site] = mod
        parent = sys.modules.get(site.rsplit('.', 1)[0])
        if parent is not None:                  # the package is already imported: `from . import x` looks at its attributes first
            setattr(parent, site.rsplit('.', 1)[1], mod)
        mods.append(mod)
    return tuple(mods)
