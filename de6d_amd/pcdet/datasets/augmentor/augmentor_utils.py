"""SlopeAug (mirror of `random_global_make_slope`, core/pcdet/datasets/augmentor/augmentor_utils.py:622-694).

The random pivot and rotation vector are a few host scalars, drawn with the same `np.random.random`
calls in the same order as the reference, so a seeded run picks the same slope.  The per-point and
per-box geometry runs on the GPU (det6d_make_slope).  Inputs may be NumPy arrays (uploaded, result
returned as NumPy) or CUDA tensors (updated in place, returned as tensors).
"""
import numpy as np
import torch

from ....ops import fused


def _uniform(n=1):
    return (np.random.random(n) - 0.5) * 2


def rotvec_to_matrix(v):
    """Rodrigues' formula (what scipy's Rotation.from_rotvec(v).as_matrix() evaluates)"""
    v = np.asarray(v, np.float64)
    theta = np.linalg.norm(v)
    if theta < 1e-12:
        return np.eye(3)
    a = v / theta
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(theta) * K + (1 - np.cos(theta)) * (K @ K)


def matrix_to_euler_XYZ(R):
    """intrinsic X-Y-Z angles (a, b, c) with R = Rx(a) Ry(b) Rz(c); |b| < pi/2 for every slope angle"""
    return np.array([np.arctan2(-R[1, 2], R[2, 2]), np.arcsin(np.clip(R[0, 2], -1.0, 1.0)), np.arctan2(-R[0, 1], R[0, 0])])


def slope_params(rotate_point, rotate_angle):
    """the 16 doubles det6d_make_slope reads: pivot | R | k | sensor side | pitch, roll increments"""
    x0, y0 = rotate_point[0], rotate_point[1]
    k = rotate_angle[1] / (rotate_angle[0] + 1e-6)
    R = rotvec_to_matrix(rotate_angle)
    euler = matrix_to_euler_XYZ(R)
    return np.concatenate([np.asarray(rotate_point, np.float64), R.reshape(-1),
                           [k, np.sign(k * (0 - x0) + y0 - 0), euler[1], euler[0]]])


def _to_device(gt_boxes, points):
    as_numpy = not torch.is_tensor(points)
    pts = torch.from_numpy(np.ascontiguousarray(points, np.float32)).cuda() if as_numpy else points
    if torch.is_tensor(gt_boxes):
        boxes = gt_boxes.double()
    else:
        boxes = torch.from_numpy(np.asarray(gt_boxes, np.float64))
    if boxes.shape[1] < 9:
        boxes = torch.cat([boxes, boxes.new_zeros((boxes.shape[0], 9 - boxes.shape[1]))], 1)
    return boxes.contiguous().to(pts.device), pts, as_numpy


def random_global_make_slope(gt_boxes, points, params=None, rotate_point=None, rotate_angle=None, smooth=False):
    assert params is not None
    dist_mean, dist_var, angle_mean, angle_var = params
    if rotate_point is None:
        polar = np.array([dist_mean, 0]) + _uniform(2) * np.array([dist_var, 0])
        rotate_point = np.array([polar[0] * np.cos(polar[1]), polar[0] * np.sin(polar[1]), 0])
    if rotate_angle is None:
        x0, y0 = rotate_point[0], rotate_point[1]
        k1 = -1 / (y0 / x0 + 1e-6)
        axis = np.array([x0 - 0, y0 - (-x0 * k1 + y0), 0])
        axis /= np.linalg.norm(axis)
        rotate_angle = axis * (angle_mean + _uniform() * angle_var)

    boxes, pts, as_numpy = _to_device(gt_boxes, points)
    if smooth:   # two half-angle slopes whose pivots lie on the arc of radius x0 / |angle| (reference :649-668)
        radius, bins = rotate_point[0] / np.abs(rotate_angle[1]), 2
        alpha, dist = rotate_angle[1], rotate_point[0]
        for theta in np.linspace(0, alpha, bins):
            pivot = np.array([dist, 0, radius]) + np.array([-radius * np.sin(theta), 0, -radius * np.cos(theta)])
            fused.make_slope(pts, boxes, slope_params(pivot, np.array([0, alpha / bins, 0])))
    else:
        fused.make_slope(pts, boxes, slope_params(rotate_point, rotate_angle))
    if as_numpy:
        return boxes.cpu().numpy(), pts.cpu().numpy(), rotate_point, rotate_angle
    return boxes, pts, rotate_point, rotate_angle
