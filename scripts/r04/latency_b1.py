"""one frame on an idle chip (bench.py's latency_b1 leg alone): captured graph, B = 1 x 16384; uniform and ray-cast scenes"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import synth_points
from de6d_amd.runtime import load_config, build_model, GraphedDet6D

cfg = load_config('kitti_models/det6d_car.yaml')
model = build_model(cfg, seed=1234, device='cuda')
n = 16384
for scene in ('uniform', 'beam'):
    one = torch.from_numpy(synth_points(4242, 1, n, tilt=False, scene=scene)).cuda()
    lat1 = GraphedDet6D(model, 1, n, points=one)
    for _ in range(3):
        lat1.launch(); lat1.finalize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        lat1.launch(); lat1.finalize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(scene, 'latency_b1 ms: min %.3f median %.3f max %.3f' % (ts[0], ts[len(ts) // 2], ts[-1]), flush=True)
    del lat1
