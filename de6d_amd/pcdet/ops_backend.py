"""Single place where the pcdet mirror binds to the compiled backend — the analogue of the
reference's `from . import pointnet2_batch_cuda as pointnet2` (pointnet2_utils.py:7) and
`from . import iou3d_nms_cuda` (iou3d_nms_utils.py:9)."""
try:
    from ..ops import fused, iou3d_nms_hip, pointnet2_batch_hip  # noqa: F401
except ImportError:  # `pcdet` imported as a top-level package (de6d_amd/ on sys.path)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from de6d_amd.ops import fused, iou3d_nms_hip, pointnet2_batch_hip  # noqa: F401
