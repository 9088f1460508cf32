"""same reader as the KITTI one (the reference's two kitti_common.py files are identical)"""
from ...kitti.kitti_object_eval_python.kitti_common import get_image_index_str, get_label_anno, get_label_annos  # noqa: F401
