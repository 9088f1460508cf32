"""Shared synthetic inputs for the parity tests (seeded; SURVEY.md 8d scene recipe, reduced)."""
import numpy as np


def make_scene(seed, n, tilt=False, dup_frac=0.05):
    """KITTI-like cloud (n,4) [x,y,z,intensity]: ground plane + boxes + clutter, with exact
    duplicate points like the reference's sample_points padding (data_processor.py:170-175)."""
    rng = np.random.default_rng(seed)
    n_ground = int(0.7 * n)
    g = np.stack([rng.uniform(0, 70.4, n_ground), rng.uniform(-40, 40, n_ground),
                  -1.7 + 0.02 * rng.standard_normal(n_ground)], 1)
    if tilt:
        x0 = rng.uniform(14, 30)
        ang = np.deg2rad(rng.uniform(10, 20))
        far = g[:, 0] > x0
        dx = g[far, 0] - x0
        g[far, 0] = x0 + dx * np.cos(ang)
        g[far, 2] = g[far, 2] + dx * np.sin(ang)
    n_obj = n - n_ground
    centers = np.stack([rng.uniform(5, 60, 40), rng.uniform(-30, 30, 40), np.full(40, -0.9)], 1)
    which = rng.integers(0, 40, n_obj)
    o = centers[which] + rng.uniform(-0.5, 0.5, (n_obj, 3)) * np.array([3.9, 1.6, 1.56])
    pts = np.concatenate([g, o], 0)
    rng.shuffle(pts)
    ndup = int(dup_frac * n)
    if ndup > 0:
        src = rng.integers(0, n, ndup)
        dst = rng.integers(0, n, ndup)
        pts[dst] = pts[src]
    inten = rng.uniform(0, 1, (n, 1))
    return np.concatenate([pts, inten], 1).astype(np.float32)


def make_batch(seed0, b, n, **kw):
    return np.stack([make_scene(seed0 + i, n, **kw) for i in range(b)], 0)


def random_boxes(seed, k, spread=30.0):
    rng = np.random.default_rng(seed)
    b = np.zeros((k, 7), np.float32)
    b[:, 0] = rng.uniform(0, spread, k)
    b[:, 1] = rng.uniform(-spread / 3, spread / 3, k)
    b[:, 2] = rng.uniform(-1, 1, k)
    b[:, 3] = rng.uniform(1, 5, k)
    b[:, 4] = rng.uniform(1, 3, k)
    b[:, 5] = rng.uniform(1, 2, k)
    b[:, 6] = rng.uniform(-4, 7, k)
    return b
