"""KITTI calibration (mirror of core/pcdet/utils/calibration_kitti.py: `get_calib_from_file`,
`Calibration` with P2 / R0 / V2C and the LiDAR -> rect -> image maps).

The per-detection conversions run on the GPU (det6d_kitti_annos); `packed()` lays the matrices out
the way that kernel reads them.  The NumPy point maps are kept for host-side data preparation (FOV
flags etc.), where the reference uses them too.
"""
import numpy as np


def get_calib_from_file(calib_file):
    """calib.txt of the KITTI object benchmark: lines P0..P3, R0_rect, Tr_velo_to_cam"""
    with open(calib_file) as f:
        rows = [ln.strip().split(' ')[1:] for ln in f.readlines()]

    def mat(i, shape):
        return np.array(rows[i], dtype=np.float32).reshape(shape)

    return {'P2': mat(2, (3, 4)), 'P3': mat(3, (3, 4)), 'R0': mat(4, (3, 3)), 'Tr_velo2cam': mat(5, (3, 4))}


class Calibration(object):
    def __init__(self, calib_file):
        calib = calib_file if isinstance(calib_file, dict) else get_calib_from_file(calib_file)
        self.P2 = np.asarray(calib['P2'], np.float32)            # 3 x 4
        self.R0 = np.asarray(calib['R0'], np.float32)            # 3 x 3
        self.V2C = np.asarray(calib['Tr_velo2cam'], np.float32)  # 3 x 4
        self.cu, self.cv = self.P2[0, 2], self.P2[1, 2]
        self.fu, self.fv = self.P2[0, 0], self.P2[1, 1]
        self.tx, self.ty = self.P2[0, 3] / (-self.fu), self.P2[1, 3] / (-self.fv)

    @staticmethod
    def cart_to_hom(pts):
        return np.hstack((pts, np.ones((pts.shape[0], 1), dtype=np.float32)))

    def lidar_to_rect_matrix(self):
        """(4, 3) float32: [x, y, z, 1] @ M = rectified camera coordinates"""
        return np.dot(self.V2C.T, self.R0.T)

    def lidar_to_rect(self, pts_lidar):
        return np.dot(self.cart_to_hom(pts_lidar), self.lidar_to_rect_matrix())

    def rect_to_img(self, pts_rect):
        hom = self.cart_to_hom(pts_rect)
        proj = np.dot(hom, self.P2.T)
        return (proj[:, 0:2].T / hom[:, 2]).T, proj[:, 2] - self.P2.T[3, 2]

    def lidar_to_img(self, pts_lidar):
        return self.rect_to_img(self.lidar_to_rect(pts_lidar))

    def packed(self, image_shape=None):
        """28 float32 for det6d_kitti_annos: M (4x3) | P2 (3x4) | image height, width | 0, 0"""
        out = np.zeros(28, np.float32)
        out[0:12] = self.lidar_to_rect_matrix().reshape(-1)
        out[12:24] = self.P2.reshape(-1)
        if image_shape is not None:
            out[24], out[25] = float(image_shape[0]), float(image_shape[1])
        return out
