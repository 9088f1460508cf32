"""NumPy / scipy restatement of the reference's SlopeAug geometry — TEST INFRASTRUCTURE ONLY.

  random_global_make_slope   core/pcdet/datasets/augmentor/augmentor_utils.py:622-694
  boxes3d_to_corners_3d      core/pcdet/utils/box_utils.py:57-71
  limit_period               core/pcdet/utils/common_utils.py:22-25

Pinned by tests/golden/slope.npz (the reference's own functions, seeded np.random).
"""
import numpy as np
from scipy.spatial.transform import Rotation


def _uniform(n=1):
    return (np.random.random(n) - 0.5) * 2


def limit_period(val, offset=0.5, period=np.pi):
    """the reference routes NumPy input through torch float32 (common_utils.check_numpy_to_torch: .float()),
    so the wrapped angles are float32 values"""
    v = np.asarray(val).astype(np.float32)
    return v - np.floor(v / np.float32(period) + np.float32(offset)) * np.float32(period)


def draw_pivot(params):
    """augmentor_utils.py:631-634 (two uniform draws)"""
    dist_mean, dist_var = params[0], params[1]
    polar = np.array([dist_mean, 0]) + _uniform(2) * np.array([dist_var, 0])
    return np.array([polar[0] * np.cos(polar[1]), polar[0] * np.sin(polar[1]), 0])


def draw_rotvec(params, pivot):
    """augmentor_utils.py:637-646 (one uniform draw): rotation axis in the ground plane, normal to the pivot ray"""
    angle_mean, angle_var = params[2], params[3]
    x0, y0 = pivot[0], pivot[1]
    k1 = -1 / (y0 / x0 + 1e-6)
    v = np.array([x0 - 0, y0 - (-x0 * k1 + y0), 0])
    v /= np.linalg.norm(v)
    v *= angle_mean + _uniform() * angle_var
    return v


def apply_slope(gt_boxes, points, pivot, rotvec):
    """augmentor_utils.py:670-694: rotate everything beyond the pivot line, extend boxes to 9-D"""
    x0, y0 = pivot[0], pivot[1]
    k = rotvec[1] / (rotvec[0] + 1e-6)
    side = np.sign(k * (0 - x0) + y0 - 0)
    rot = Rotation.from_rotvec(rotvec).as_matrix()
    beyond = np.sign(k * (points[:, 0] - x0) + y0 - points[:, 1]) != side
    moved = points[beyond]
    moved[:, 0:3] -= pivot
    moved[:, 0:3] = moved[:, 0:3].dot(rot.T)
    moved[:, 0:3] += pivot
    points[beyond] = moved
    if gt_boxes.shape[1] < 9:
        gt_boxes = np.concatenate((gt_boxes, np.zeros([gt_boxes.shape[0], 2])), axis=1)
    beyond = np.sign(k * (gt_boxes[:, 0] - x0) + y0 - gt_boxes[:, 1]) != side
    moved = gt_boxes[beyond]
    moved[:, :3] -= pivot
    moved[:, :3] = moved[:, :3].dot(rot.T)
    moved[:, :3] += pivot
    gt_boxes[beyond] = moved
    euler = Rotation.from_rotvec(rotvec).as_euler('XYZ')
    gt_boxes[beyond, 7] += euler[1]
    gt_boxes[beyond, 8] += euler[0]
    gt_boxes[:, 6:9] = limit_period(gt_boxes[:, 6:9], offset=0.5, period=2 * np.pi)
    return gt_boxes, points


def random_global_make_slope(gt_boxes, points, params=None, rotate_point=None, rotate_angle=None, smooth=False):
    if rotate_point is None:
        rotate_point = draw_pivot(params)
    if rotate_angle is None:
        rotate_angle = draw_rotvec(params, rotate_point)
    if not smooth:
        gt_boxes, points = apply_slope(gt_boxes, points, rotate_point, rotate_angle)
        return gt_boxes, points, rotate_point, rotate_angle
    radius, bins = rotate_point[0] / np.abs(rotate_angle[1]), 2
    alpha, dist = rotate_angle[1], rotate_point[0]
    for theta in np.linspace(0, alpha, bins):
        centre = np.array([dist, 0, radius])
        pivot = centre + np.array([-radius * np.sin(theta), 0, -radius * np.cos(theta)])
        gt_boxes, points = apply_slope(gt_boxes, points, pivot, np.array([0, alpha / bins, 0]))
    return gt_boxes, points, rotate_point, rotate_angle


def boxes9_corners(boxes):
    template = np.array([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                         [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]]) / 2
    corners = boxes[:, None, 3:6].repeat(8, 1) * template[None, :, :]
    rot = Rotation.from_euler('zyx', boxes[:, 6:9]).as_matrix()
    corners[:, :, 0:3] = np.matmul(corners[:, :, 0:3], rot.transpose((0, 2, 1)))
    corners += boxes[:, None, 0:3]
    return corners
