"""SlopedKITTI evaluation entry points (core/pcdet/datasets/slopedkitti/kitti_object_eval_python/eval.py):
the shared implementation lives next to the KITTI one."""
from ...kitti.kitti_object_eval_python.eval import (  # noqa: F401
    clean_data, do_eval_slopedkitti, eval_class, get_mAP, get_mAP_R40, get_ods, get_slopedkitti_eval_result,
    get_thresholds, get_tp_score)
