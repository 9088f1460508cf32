"""Det6D detector (core/pcdet/models/detectors/det6d.py:4-30): backbone_3d -> point_head ->
post_processing.  Inference only; the training loss branch (:14-30) is out of scope."""
from ...ops_backend import fused
from .detector3d_template import Detector3DTemplate


class Det6D(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    def forward(self, batch_dict):
        if self.training:
            raise NotImplementedError('Det6D on HIP is an inference engine: call .eval() (training is out of scope)')
        for module in self.module_list:
            batch_dict = module(batch_dict)
        out = self.post_processing(batch_dict)
        if fused.PENDING_FPS_STATUS:       # eager launches of the cooperative 32768 / 65536-point sampler: did one give up?
            fused.check_fps_status()       # (post_processing has synchronised already)
        return out

    def forward_async(self, batch_dict):
        """enqueue one full pass on the current stream without blocking the host; pair with
        finalize().  Needs the fused post-processing route (class-agnostic nms_gpu, <= 512 boxes)."""
        for module in self.module_list:
            batch_dict = module(batch_dict)
        return self.post_processing_async(batch_dict)
