from .det6d import Det6D
from .detector3d_template import Detector3DTemplate

# registry by name, core/pcdet/models/detectors/__init__.py:17-31 (Det6D path only)
__all__ = {
    'Detector3DTemplate': Detector3DTemplate,
    'Det6D': Det6D,
}


def build_detector(model_cfg, num_class, dataset):
    return __all__[model_cfg.NAME](model_cfg=model_cfg, num_class=num_class, dataset=dataset)
