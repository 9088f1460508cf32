"""Seeded synthetic inputs (there is no KITTI in the snapshot: SURVEY.md 8d).  Two scene generators:

* `make_scene`   — the SURVEY.md 8d recipe: uniform ground plane + 40 car-sized boxes, 5 % exact duplicates
                   (what the reference's `sample_points` padding produces, data_processor.py:170-175);
* `beam_scene`   — a ray-cast 64-ring spinning LiDAR (HDL-64E geometry: +2 .. -24.8 deg, 0.09 deg azimuth
                   steps, sensor 1.73 m above the ground) over ground + cars + walls, cropped to the KITTI
                   range and brought to exactly n points by the reference's near / far rule
                   (data_processor.py:145-178).  Its density falls with range like a real frame's, so the
                   fill of the ball queries (and with it the compact-row gain) is realistic.

Host-side numpy only; bench.py, the tests and smoke() share it."""
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


def _ray_boxes(origin, dirs, centers, dims, yaw):
    """nearest hit distance of rays (R,3) against K yawed boxes (slab test in the box frame); inf = miss"""
    best = np.full(dirs.shape[0], np.inf)
    for c, d, a in zip(centers, dims, yaw):
        ca, sa = np.cos(a), np.sin(a)
        rot = np.array([[ca, sa, 0.0], [-sa, ca, 0.0], [0.0, 0.0, 1.0]])
        o = rot @ (origin - c)
        dl = dirs @ rot.T
        with np.errstate(divide='ignore', invalid='ignore'):
            t1 = (-d / 2 - o) / dl
            t2 = (d / 2 - o) / dl
        tn = np.nanmax(np.minimum(t1, t2), axis=1)
        tf = np.nanmin(np.maximum(t1, t2), axis=1)
        hit = (tn <= tf) & (tf > 0) & (tn > 0)
        best = np.where(hit & (tn < best), tn, best)
    return best


def sample_points_rule(pts, n, rng):
    """the reference's `sample_points` (data_processor.py:145-178): more than n points -> every far point
    (depth >= 40 m) + a random subset of the near ones; fewer -> duplicate padding; then a shuffle"""
    if len(pts) > n:
        depth = np.linalg.norm(pts[:, :3], axis=1)
        near = np.where(depth < 40.0)[0]
        far = np.where(depth >= 40.0)[0]
        if len(far) > n:
            choice = rng.choice(np.arange(len(pts)), n, replace=False)
        else:
            choice = np.concatenate([rng.choice(near, n - len(far), replace=False), far]) if len(far) else \
                rng.choice(np.arange(len(pts)), n, replace=False)
    else:
        choice = np.arange(len(pts))
        if len(pts) < n:
            extra = rng.choice(choice, n - len(pts), replace=(n - len(pts)) > len(pts))
            choice = np.concatenate([choice, extra])
    rng.shuffle(choice)
    return pts[choice]


def beam_scene(seed, n, tilt=False, rings=64, az_step_deg=0.09, fov_deg=45.0, dropout=0.1):
    """ray-cast 64-ring LiDAR frame, cropped to [0, 70.4] x [-40, 40] and sampled to n points (n,4)"""
    rng = np.random.default_rng(seed)
    sensor_h = 1.73
    elev = np.deg2rad(np.linspace(2.0, -24.8, rings))
    az = np.deg2rad(np.arange(-fov_deg, fov_deg, az_step_deg))
    e, a = np.meshgrid(elev, az, indexing='ij')
    e = e.ravel() + np.deg2rad(0.01) * rng.standard_normal(e.size)
    a = a.ravel() + np.deg2rad(0.01) * rng.standard_normal(a.size)
    dirs = np.stack([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)], 1)
    origin = np.zeros(3)
    # ground: z = -sensor_h, beyond x0 optionally a slope (SlopeAug-like ramp)
    with np.errstate(divide='ignore'):
        t_ground = np.where(dirs[:, 2] < 0, -sensor_h / dirs[:, 2], np.inf)
    if tilt:
        x0 = rng.uniform(14, 30)
        ang = np.deg2rad(rng.uniform(10, 20))
        nrm = np.array([-np.sin(ang), 0.0, np.cos(ang)])          # plane through (x0, *, -h) tilted about y
        p0 = np.array([x0, 0.0, -sensor_h])
        den = dirs @ nrm
        with np.errstate(divide='ignore', invalid='ignore'):
            t_ramp = np.where(np.abs(den) > 1e-9, (p0 @ nrm) / den, np.inf)
        t_ramp = np.where(t_ramp > 0, t_ramp, np.inf)
        flat_hit_x = t_ground * dirs[:, 0]
        t_ground = np.where(flat_hit_x <= x0, t_ground, np.where((t_ramp * dirs[:, 0]) > x0, t_ramp, np.inf))
    k = 40
    centers = np.stack([rng.uniform(5, 65, k), rng.uniform(-30, 30, k), np.full(k, -sensor_h + 0.78)], 1)
    dims = np.tile(np.array([3.9, 1.6, 1.56]), (k, 1)) * rng.uniform(0.9, 1.1, (k, 3))
    yaw = rng.uniform(-np.pi, np.pi, k)
    nw = 24                                                       # facades / hedges: what returns the far beams
    walls_c = np.stack([rng.uniform(10, 70, nw), rng.uniform(6, 38, nw) * rng.choice([-1.0, 1.0], nw), np.full(nw, 1.0)], 1)
    walls_d = np.stack([rng.uniform(8, 40, nw), np.full(nw, 0.3), np.full(nw, 6.0)], 1)
    walls_y = np.deg2rad(rng.uniform(-25, 25, nw))                # roughly along the road
    t_obj = _ray_boxes(origin, dirs, np.concatenate([centers, walls_c]), np.concatenate([dims, walls_d]),
                       np.concatenate([yaw, walls_y]))
    t = np.minimum(t_ground, t_obj)
    ok = np.isfinite(t) & (t < 80.0) & (rng.uniform(0, 1, t.size) >= dropout)
    t = t[ok] + 0.02 * rng.standard_normal(ok.sum())
    pts = dirs[ok] * t[:, None]
    keep = (pts[:, 0] >= 0) & (pts[:, 0] <= 70.4) & (pts[:, 1] >= -40) & (pts[:, 1] <= 40)
    pts = pts[keep]
    inten = rng.uniform(0, 1, (len(pts), 1))
    cloud = np.concatenate([pts, inten], 1).astype(np.float32)
    return sample_points_rule(cloud, n, rng).astype(np.float32)


def beam_batch(seed0, b, n, **kw):
    return np.stack([beam_scene(seed0 + i, n, **kw) for i in range(b)], 0)


def points_tensor(batch):
    """(B, N, 4) scenes -> (B*N, 5) rows [batch index, x, y, z, intensity] (collate_batch, dataset.py:171-176)"""
    b, n, _ = batch.shape
    bidx = np.repeat(np.arange(b, dtype=np.float32), n)[:, None]
    return np.concatenate([bidx, batch.reshape(b * n, 4)], 1).astype(np.float32)
