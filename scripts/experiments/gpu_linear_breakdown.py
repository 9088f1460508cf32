import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from de6d_amd.runtime import load_config, build_model
from de6d_amd.ops import fused
from bench import synth_points
cfg = load_config('kitti_models/det6d_car.yaml')
model = build_model(cfg, device='cuda')
B, N = 8, 16384
points = torch.from_numpy(synth_points(1000, B, N)).cuda()
with torch.no_grad():
    for _ in range(3): model({'batch_size': B, 'points': points})
    acc = None
    for it in range(5):
        fused.LINEAR_EVENTS = []
        model({'batch_size': B, 'points': points}); torch.cuda.synchronize()
        ev = fused.LINEAR_EVENTS; fused.LINEAR_EVENTS = None
        ms = [e0.elapsed_time(e1) for e0, e1, *_ in ev]
        acc = ms if acc is None else [min(a, b) for a, b in zip(acc, ms)]
tot = 0
for (e0, e1, r, k, n), t in zip(ev, acc):
    cap = None
    if torch.is_tensor(r):
        h = r.cpu().numpy(); r, cap = int(h[0]), (int(h[7]), int(h[8]), int(h[9]))
    fl = 2.0 * r * k * n
    tot += t
    print("rows %8d K %4d N %4d  %8.1f us  %6.1f TF  %6.2f GF  %s" % (r, k, n, t * 1e3, fl / t / 1e9, fl / 1e9, cap or ''))
print("total ms", tot)
