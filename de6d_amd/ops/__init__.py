"""Drop-in op modules.

``pointnet2_batch_hip`` / ``iou3d_nms_hip`` expose exactly the function names and argument orders
of the reference's pybind modules ``pointnet2_batch_cuda`` / ``iou3d_nms_cuda``
(core/pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:11-30,
core/pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17), so the reference's own
``pointnet2_utils.py`` / ``iou3d_nms_utils.py`` run unmodified on top of them.
"""
