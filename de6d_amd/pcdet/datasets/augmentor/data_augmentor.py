"""DataAugmentor restricted to the augmentation this repo's path adds: `random_make_slope_in_scene`
(mirror of core/pcdet/datasets/augmentor/data_augmentor.py:265-282; YAML keys SLOPE_DISTANCE,
SLOPE_ANGLE, SMOOTH, PROB).  The other augmentors of the reference are training-side and out of scope;
naming them raises NotImplementedError."""
from functools import partial

import numpy as np

from . import augmentor_utils


class DataAugmentor(object):
    def __init__(self, root_path, augmentor_configs, class_names, logger=None):
        self.root_path, self.class_names, self.logger = root_path, class_names, logger
        self.data_augmentor_queue = []
        cfg_list = augmentor_configs if isinstance(augmentor_configs, list) else augmentor_configs.AUG_CONFIG_LIST
        disabled = [] if isinstance(augmentor_configs, list) else augmentor_configs.get('DISABLE_AUG_LIST', [])
        for cur_cfg in cfg_list:
            if cur_cfg['NAME'] in disabled:
                continue
            if cur_cfg['NAME'] != 'random_make_slope_in_scene':
                raise NotImplementedError('%s: only random_make_slope_in_scene is on this path' % cur_cfg['NAME'])
            self.data_augmentor_queue.append(self.random_make_slope_in_scene(config=cur_cfg))

    def random_make_slope_in_scene(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.random_make_slope_in_scene, config=config)
        dist_mean, dist_var = config['SLOPE_DISTANCE']['MEAN'], config['SLOPE_DISTANCE']['VAR']
        angle_mean, angle_var = np.deg2rad([config['SLOPE_ANGLE']['MEAN'], config['SLOPE_ANGLE']['VAR']])
        choice = np.random.random()
        gt_boxes, points = data_dict['gt_boxes'], data_dict['points']
        gt_boxes = np.concatenate((gt_boxes, np.zeros([gt_boxes.shape[0], 2])), axis=1)   # 9-D boxes either way
        if choice < config['PROB']:
            gt_boxes, points, *_ = augmentor_utils.random_global_make_slope(
                gt_boxes, points, params=(dist_mean, dist_var, angle_mean, angle_var), smooth=config.get('SMOOTH', False))
        data_dict['gt_boxes'], data_dict['points'] = gt_boxes, points
        return data_dict

    def forward(self, data_dict):
        for cur_augmentor in self.data_augmentor_queue:
            data_dict = cur_augmentor(data_dict=data_dict)
        return data_dict
