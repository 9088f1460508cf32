"""NumPy restatement of the reference's prediction-dict generation — TEST INFRASTRUCTURE ONLY.

Follows, array operation by array operation, what the reference does on the host after the model
(SURVEY.md §8 f2):

  Calibration                           core/pcdet/utils/calibration_kitti.py:23-83
  boxes3d_lidar_to_kitti_camera         core/pcdet/utils/box_utils.py:196-212
  boxes3d_to_corners3d_kitti_camera     box_utils.py:215-258
  boxes3d_kitti_camera_to_imageboxes    box_utils.py:261-281
  generate_prediction_dicts             core/pcdet/datasets/kitti/kitti_dataset.py:277-351
                                        core/pcdet/datasets/slopedkitti/kitti_dataset.py:299-379 (pitch, roll)

Pinned by tests/golden/annos.npz, produced by the reference's own functions
(tests/golden/make_golden.py: gen_annos).
"""
import numpy as np

F32 = np.float32


class Calib(object):
    """P2 (3,4), R0 (3,3), Tr_velo2cam (3,4), all float32"""

    def __init__(self, P2, R0, V2C):
        self.P2, self.R0, self.V2C = (np.asarray(a, F32) for a in (P2, R0, V2C))

    def lidar_to_rect(self, pts):
        hom = np.hstack((pts, np.ones((pts.shape[0], 1), dtype=F32)))
        return np.dot(hom, np.dot(self.V2C.T, self.R0.T))

    def rect_to_img(self, pts):
        hom = np.hstack((pts, np.ones((pts.shape[0], 1), dtype=F32)))
        proj = np.dot(hom, self.P2.T)
        return (proj[:, 0:2].T / hom[:, 2]).T


def lidar_to_camera_boxes(boxes, calib):
    b = np.array(boxes, copy=True)
    xyz, l, w, h, r = b[:, 0:3], b[:, 3:4], b[:, 4:5], b[:, 5:6], b[:, 6:7]
    xyz[:, 2] -= h.reshape(-1) / 2
    return np.concatenate([calib.lidar_to_rect(xyz), l, h, w, -r - np.pi / 2], axis=-1)


def camera_corners(cam):
    n = cam.shape[0]
    l, h, w = cam[:, 3], cam[:, 4], cam[:, 5]
    xs = np.array([l / 2., l / 2., -l / 2., -l / 2., l / 2., l / 2., -l / 2., -l / 2], dtype=F32).T
    zs = np.array([w / 2., -w / 2., -w / 2., w / 2., w / 2., -w / 2., -w / 2., w / 2.], dtype=F32).T
    ys = np.zeros((n, 8), dtype=F32)
    ys[:, 4:8] = -h.reshape(n, 1).repeat(4, axis=1)
    ry = cam[:, 6]
    zeros, ones = np.zeros(ry.size, dtype=F32), np.ones(ry.size, dtype=F32)
    rot = np.transpose(np.array([[np.cos(ry), zeros, -np.sin(ry)], [zeros, ones, zeros], [np.sin(ry), zeros, np.cos(ry)]]),
                       (2, 0, 1))
    local = np.concatenate((xs.reshape(-1, 8, 1), ys.reshape(-1, 8, 1), zs.reshape(-1, 8, 1)), axis=2)
    turned = np.matmul(local, rot)
    out = turned + cam[:, None, 0:3]
    return out.astype(F32)


def camera_to_image_boxes(cam, calib, image_shape=None):
    uv = calib.rect_to_img(camera_corners(cam).reshape(-1, 3)).reshape(-1, 8, 2)
    box = np.concatenate([np.min(uv, axis=1), np.max(uv, axis=1)], axis=1)
    if image_shape is not None:
        box[:, 0] = np.clip(box[:, 0], a_min=0, a_max=image_shape[1] - 1)
        box[:, 1] = np.clip(box[:, 1], a_min=0, a_max=image_shape[0] - 1)
        box[:, 2] = np.clip(box[:, 2], a_min=0, a_max=image_shape[1] - 1)
        box[:, 3] = np.clip(box[:, 3], a_min=0, a_max=image_shape[0] - 1)
    return box


def prediction_dict(boxes, scores, labels, calib, image_shape, class_names, sloped=False):
    """one frame: numpy boxes (K, 7|9), scores (K), labels (K) int -> annotation dict"""
    k = scores.shape[0]
    d = {'name': np.zeros(k), 'truncated': np.zeros(k), 'occluded': np.zeros(k), 'alpha': np.zeros(k),
         'bbox': np.zeros([k, 4]), 'dimensions': np.zeros([k, 3]), 'location': np.zeros([k, 3]),
         'rotation_y': np.zeros(k), 'score': np.zeros(k), 'boxes_lidar': np.zeros([k, 7])}
    if sloped:
        d['pitch'], d['roll'] = np.zeros(k), np.zeros(k)
    if k == 0:
        return d
    cam = lidar_to_camera_boxes(boxes, calib)
    d['name'] = np.array(class_names)[labels - 1]
    d['alpha'] = -np.arctan2(-boxes[:, 1], boxes[:, 0]) + cam[:, 6]
    d['bbox'] = camera_to_image_boxes(cam, calib, image_shape=image_shape)
    d['dimensions'], d['location'], d['rotation_y'] = cam[:, 3:6], cam[:, 0:3], cam[:, 6]
    if sloped and boxes.shape[1] >= 9:
        d['pitch'], d['roll'] = boxes[:, 7], boxes[:, 8]
    d['score'], d['boxes_lidar'] = scores, boxes
    return d


def kitti_lines(d, sloped=False):
    """the label-file lines of one frame (kitti_dataset.py:337-349; the sloped variant adds pitch, roll)"""
    lines = []
    for i in range(len(d['bbox'])):
        vals = [d['alpha'][i], *d['bbox'][i], d['dimensions'][i][1], d['dimensions'][i][2], d['dimensions'][i][0],
                *d['location'][i], d['rotation_y'][i]]
        if sloped:
            vals += [d['pitch'][i], d['roll'][i]]
        vals.append(d['score'][i])
        lines.append('%s -1 -1 ' % d['name'][i] + ' '.join('%.4f' % v for v in vals))
    return lines
