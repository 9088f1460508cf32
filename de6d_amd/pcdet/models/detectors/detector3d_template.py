"""Detector3DTemplate — module registry, topology and post-processing of
core/pcdet/models/detectors/detector3d_template.py (:14-50 build, :68-83/:141-158 builders,
:178-284 post_processing, :330-411 checkpoint loading) for the point-based Det6D path.

post_processing runs as ONE HIP kernel launch for the whole batch (csrc/iou3d_nms.hip:
postprocess_kernel: sigmoid, class max, score filter, stable sort, rotated NMS, selection) and
one device->host copy of the per-scene counts, instead of the reference's per-scene Python loop
with two sorts and a blocking NMS."""
import os

import torch
import torch.nn as nn

from .. import backbones_3d, dense_heads
from ..model_utils import model_nms_utils
from ...ops_backend import fused


class Detector3DTemplate(nn.Module):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.dataset = dataset
        self.class_names = dataset.class_names
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        self.module_topology = ['vfe', 'backbone_3d', 'map_to_bev_module', 'pfe', 'backbone_2d', 'dense_head',
                                'point_head', 'roi_head']
        # Conv+BN are folded into device matrices at first use.  Any load_state_dict() (also the plain nn.Module one)
        # drops them; `weights_version` lets captured graphs (runtime.GraphedDet6D), which hold pointers to the folded
        # tensors, refuse to replay after the weights changed.
        self.weights_version = 0
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module._invalidate_folded())

    @property
    def mode(self):
        return 'TRAIN' if self.training else 'TEST'

    def update_global_step(self):
        self.global_step += 1

    def build_networks(self):
        info = {
            'module_list': [],
            'num_rawpoint_features': self.dataset.point_feature_encoder.num_point_features,
            'num_point_features': self.dataset.point_feature_encoder.num_point_features,
            'grid_size': self.dataset.grid_size,
            'point_cloud_range': self.dataset.point_cloud_range,
            'voxel_size': self.dataset.voxel_size,
            'depth_downsample_factor': self.dataset.depth_downsample_factor,
        }
        for name in self.module_topology:
            module, info = getattr(self, 'build_%s' % name)(model_info_dict=info)
            self.add_module(name, module)
        return info['module_list']

    def _absent(self, key, what):
        if self.model_cfg.get(key, None) is not None:
            raise NotImplementedError('%s (%s) is outside the Det6D hot path (SURVEY.md section 8)' % (key, what))

    def build_vfe(self, model_info_dict):
        self._absent('VFE', 'voxel feature encoders')
        return None, model_info_dict

    def build_map_to_bev_module(self, model_info_dict):
        self._absent('MAP_TO_BEV', 'BEV scatter')
        return None, model_info_dict

    def build_pfe(self, model_info_dict):
        self._absent('PFE', 'voxel set abstraction')
        return None, model_info_dict

    def build_backbone_2d(self, model_info_dict):
        self._absent('BACKBONE_2D', '2-D backbones')
        return None, model_info_dict

    def build_dense_head(self, model_info_dict):
        self._absent('DENSE_HEAD', 'anchor/center heads')
        return None, model_info_dict

    def build_roi_head(self, model_info_dict):
        self._absent('ROI_HEAD', 'second-stage heads')
        return None, model_info_dict

    def build_backbone_3d(self, model_info_dict):
        cfg = self.model_cfg.get('BACKBONE_3D', None)
        if cfg is None:
            return None, model_info_dict
        module = backbones_3d.__all__[cfg.NAME](
            model_cfg=cfg, input_channels=model_info_dict['num_point_features'],
            grid_size=model_info_dict['grid_size'], voxel_size=model_info_dict['voxel_size'],
            point_cloud_range=model_info_dict['point_cloud_range'])
        model_info_dict['module_list'].append(module)
        model_info_dict['num_point_features'] = module.num_point_features
        return module, model_info_dict

    def build_point_head(self, model_info_dict):
        cfg = self.model_cfg.get('POINT_HEAD', None)
        if cfg is None:
            return None, model_info_dict
        if cfg.get('USE_POINT_FEATURES_BEFORE_FUSION', False):
            num_point_features = model_info_dict['num_point_features_before_fusion']
        else:
            num_point_features = model_info_dict['num_point_features']
        module = dense_heads.__all__[cfg.NAME](
            model_cfg=cfg, input_channels=num_point_features,
            num_class=self.num_class if not cfg.CLASS_AGNOSTIC else 1,
            predict_boxes_when_training=self.model_cfg.get('ROI_HEAD', False))
        model_info_dict['module_list'].append(module)
        return module, model_info_dict

    def forward(self, **kwargs):
        raise NotImplementedError

    # ------------------------------------------------------------------ post-processing
    def post_processing(self, batch_dict):
        cfg = self.model_cfg.POST_PROCESSING
        nms_cfg = cfg.NMS_CONFIG
        batch_size = batch_dict['batch_size']
        box_preds = batch_dict['batch_box_preds']
        cls_preds = batch_dict['batch_cls_preds']
        fast = (not nms_cfg.MULTI_CLASSES_NMS and nms_cfg.NMS_TYPE == 'nms_gpu'
                and not batch_dict['cls_preds_normalized'] and not isinstance(cls_preds, list)
                and box_preds.dim() == 2 and box_preds.shape[1] == 9 and box_preds.shape[0] % batch_size == 0
                and box_preds.shape[0] // batch_size <= 1024 and not cfg.get('OUTPUT_RAW_SCORE', False)
                and not batch_dict.get('has_class_labels', False))
        if nms_cfg.MULTI_CLASSES_NMS:
            raise NotImplementedError('MULTI_CLASSES_NMS is not used by Det6D configs')
        pred_dicts = []
        if fast:
            handle = self.post_processing_async(batch_dict)
            pred_dicts = self.finalize(handle)
        else:  # generic per-scene route through the op-level API (same semantics)
            for i in range(batch_size):
                if batch_dict.get('batch_index', None) is not None:
                    mask = batch_dict['batch_index'] == i
                else:
                    mask = i
                bp = box_preds[mask]
                cp = cls_preds[mask]
                if not batch_dict['cls_preds_normalized']:
                    cp = fused.sigmoid_pow(cp.contiguous(), 1.0)   # the library's deterministic sigmoid
                cp, lp = torch.max(cp, dim=-1)
                lp = batch_dict['roi_labels'][i] if batch_dict.get('has_class_labels', False) else lp + 1
                sel, sel_scores = model_nms_utils.class_agnostic_nms(cp, bp, nms_cfg, cfg.SCORE_THRESH)
                pred_dicts.append({'pred_boxes': bp[sel], 'pred_scores': sel_scores, 'pred_labels': lp[sel]})
        recall_dict = {}
        if 'gt_boxes' in batch_dict:
            for i, pd in enumerate(pred_dicts):
                recall_dict = self.generate_recall_record(pd['pred_boxes'], recall_dict, i, batch_dict,
                                                          cfg.RECALL_THRESH_LIST)
        return pred_dicts, recall_dict

    def post_processing_async(self, batch_dict):
        """launch the fused post-processing kernel and an async copy of the per-scene counts;
        nothing here blocks the host (finalize() does)"""
        cfg = self.model_cfg.POST_PROCESSING
        nms_cfg = cfg.NMS_CONFIG
        boxes, scores, labels, index, count = fused.postprocess(
            batch_dict['batch_cls_preds'].contiguous(), batch_dict['batch_box_preds'].contiguous(),
            batch_dict['batch_size'], cfg.SCORE_THRESH, nms_cfg.NMS_PRE_MAXSIZE, nms_cfg.NMS_POST_MAXSIZE,
            nms_cfg.NMS_THRESH)
        count_host = torch.empty(count.shape, dtype=count.dtype, pin_memory=True)
        count_host.copy_(count, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return dict(boxes=boxes, scores=scores, labels=labels, index=index, count=count, count_host=count_host,
                    done=done)

    @staticmethod
    def finalize(handle):
        """wait for one forward pass and slice its detections: the only host sync of a pass"""
        handle['done'].synchronize()
        out = []
        for i, k in enumerate(handle['count_host'].tolist()):
            out.append({'pred_boxes': handle['boxes'][i, :k], 'pred_scores': handle['scores'][i, :k],
                        'pred_labels': handle['labels'][i, :k].long()})
        return out

    @staticmethod
    def generate_recall_record(box_preds, recall_dict, batch_index, data_dict=None, thresh_list=None):
        """recall bookkeeping of detector3d_template.py:286-328 on the first 7 box dims"""
        from ...ops.iou3d_nms import iou3d_nms_utils
        if 'gt_boxes' not in data_dict:
            return recall_dict
        rois = data_dict['rois'][batch_index] if 'rois' in data_dict else None
        gt_boxes = data_dict['gt_boxes'][batch_index]
        if len(recall_dict) == 0:
            recall_dict = {'gt': 0}
            for t in thresh_list:
                recall_dict['roi_%s' % str(t)] = 0
                recall_dict['rcnn_%s' % str(t)] = 0
        k = len(gt_boxes) - 1
        while k >= 0 and gt_boxes[k].sum() == 0:
            k -= 1
        cur_gt = gt_boxes[:k + 1]
        if cur_gt.shape[0] > 0:
            if box_preds.shape[0] > 0:
                iou3d_rcnn = iou3d_nms_utils.boxes_iou3d_gpu(box_preds[:, 0:7].contiguous(), cur_gt[:, 0:7].contiguous())
            else:
                iou3d_rcnn = torch.zeros((0, cur_gt.shape[0]), device=cur_gt.device)
            iou3d_roi = None
            if rois is not None:
                iou3d_roi = iou3d_nms_utils.boxes_iou3d_gpu(rois[:, 0:7].contiguous(), cur_gt[:, 0:7].contiguous())
            for t in thresh_list:
                if iou3d_rcnn.shape[0] > 0:
                    recall_dict['rcnn_%s' % str(t)] += (iou3d_rcnn.max(dim=0)[0] > t).sum().item()
                if iou3d_roi is not None:
                    recall_dict['roi_%s' % str(t)] += (iou3d_roi.max(dim=0)[0] > t).sum().item()
            recall_dict['gt'] += cur_gt.shape[0]
        return recall_dict

    # ------------------------------------------------------------------ checkpoints
    def invalidate_folded(self):
        """Call after editing parameters or BatchNorm statistics IN PLACE (param.data.copy_, an EMA swap, ...): the folded
        Conv+BN matrices the HIP path multiplies with are cached per module and re-folded only by train() / eval() mode
        CHANGES, load_state_dict-style loaders and this call; captured passes (runtime.GraphedDet6D) refuse to replay
        across it (`weights_version`)."""
        self._invalidate_folded()

    def _invalidate_folded(self):
        self.weights_version += 1
        for m in self.modules():
            if hasattr(m, 'invalidate') and m is not self:
                m.invalidate()

    def train(self, mode=True):
        if mode != self.training:
            self._invalidate_folded()
        return super().train(mode)

    def _load_state_dict(self, model_state_disk, *, strict=True):
        state_dict = self.state_dict()
        update = {k: v for k, v in model_state_disk.items()
                  if k in state_dict and state_dict[k].shape == v.shape}
        if strict:
            self.load_state_dict(update)
        else:
            state_dict.update(update)
            self.load_state_dict(state_dict)
        self._invalidate_folded()
        return state_dict, update

    def load_params_from_file(self, filename, logger, to_cpu=False):
        if not os.path.isfile(filename):
            raise FileNotFoundError
        logger.info('==> Loading parameters from checkpoint %s to %s' % (filename, 'CPU' if to_cpu else 'GPU'))
        checkpoint = torch.load(filename, map_location=torch.device('cpu') if to_cpu else None, weights_only=False)
        version = checkpoint.get('version', None)
        if version is not None:
            logger.info('==> Checkpoint trained from version: %s' % version)
        state_dict, update = self._load_state_dict(checkpoint['model_state'], strict=False)
        for key in state_dict:
            if key not in update:
                logger.info('Not updated weight %s: %s' % (key, str(state_dict[key].shape)))
        logger.info('==> Done (loaded %d/%d)' % (len(update), len(state_dict)))
