"""Label-file reader for the evaluator (the part of core/pcdet/datasets/kitti/kitti_object_eval_python/
kitti_common.py the evaluation path uses: get_label_anno :294-330, get_label_annos :332-347).

One KITTI label line is `name truncated occluded alpha x1 y1 x2 y2 h w l x y z ry [score]`; the files this
repo's SlopedKittiDataset writes carry `pitch roll` before the score (slopedkitti/kitti_dataset.py:365-376)
and are read back too.  `dimensions` are returned in the camera order l, h, w like the reference does.
"""
import pathlib
import re

import numpy as np


def get_label_anno(label_path):
    rows = [ln.strip().split(' ') for ln in open(label_path, 'r').readlines() if ln.strip()]
    n = len(rows)
    width = len(rows[0]) if n else 15
    sloped = width in (17, 18)
    num = np.array([[float(v) for v in r[1:]] for r in rows], np.float64).reshape(n, max(width - 1, 14))
    anno = {
        'name': np.array([r[0] for r in rows]),
        'truncated': num[:, 0],
        'occluded': num[:, 1].astype(np.int64),
        'alpha': num[:, 2],
        'bbox': num[:, 3:7].reshape(-1, 4),
        'dimensions': num[:, 7:10].reshape(-1, 3)[:, [2, 0, 1]],   # file order h, w, l -> l, h, w
        'location': num[:, 10:13].reshape(-1, 3),
        'rotation_y': num[:, 13].reshape(-1),
    }
    if sloped:
        anno['pitch'], anno['roll'] = num[:, 14], num[:, 15]
    has_score = width in (16, 18)
    anno['score'] = num[:, -1].copy() if has_score else np.zeros([n])
    return anno


def get_image_index_str(img_idx):
    return '%06d' % img_idx


def get_label_annos(label_folder, image_ids=None):
    folder = pathlib.Path(label_folder)
    if image_ids is None:
        pattern = re.compile(r'^\d{6}.txt$')
        image_ids = sorted(int(p.stem) for p in folder.glob('*.txt') if pattern.match(p.name))
    if not isinstance(image_ids, list):
        image_ids = list(range(image_ids))
    return [get_label_anno(folder / (get_image_index_str(i) + '.txt')) for i in image_ids]
