"""shared helpers of the evaluator tests: the annotations stored in tests/golden/kitti_eval.npz"""
import os

import numpy as np

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'kitti_eval.npz'))
N_FRAMES = int(GOLD['n_frames'])
KEYS = ('name', 'truncated', 'occluded', 'alpha', 'bbox', 'dimensions', 'location', 'rotation_y', 'pitch', 'roll', 'score')
CLASSES = ['Car', 'Pedestrian', 'Cyclist']


def annos(kind):
    return [{k: GOLD['%s_%d_%s' % (kind, f, k)].copy() for k in KEYS} for f in range(N_FRAMES)]
