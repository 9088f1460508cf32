"""times det6d_prepare_points on KITTI-sized raw frames (events on the launch stream)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from de6d_amd.ops import fused
from de6d_amd.pcdet.datasets import collate_frames

def lidar_frame(seed, n):
    rng = np.random.default_rng(seed)
    r = rng.gamma(2.0, 12.0, n); a = rng.uniform(-np.pi, np.pi, n)
    return np.stack([r * np.cos(a), r * np.sin(a), rng.normal(-1.2, 0.6, n), rng.uniform(0, 1, n)], 1).astype(np.float32)

RANGE = [0, -40, -3, 70.4, 40, 1]
for b, n_raw in ((8, 120000), (64, 120000), (8, 30000)):
    frames = [lidar_frame(i, n_raw) for i in range(b)]
    raw, offsets, _ = collate_frames(frames)
    for _ in range(3): out, n_in = fused.prepare_points(raw, offsets, RANGE, 16384, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps): out, n_in = fused.prepare_points(raw, offsets, RANGE, 16384, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    bytes_alg = b * (n_raw * 16 + 16384 * 20)
    print("B %3d raw %6d: %.3f ms  %.0f scenes/s  %.1f GB/s algorithmic (in-range %d)" % (b, n_raw, ms, b / ms * 1e3, bytes_alg / ms / 1e6, int(n_in[0])))
