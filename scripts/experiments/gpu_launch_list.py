"""list the kernel launches of ONE captured pass, in order (run under rocprofv3 --kernel-trace)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from de6d_amd.runtime import load_config, build_model, Det6DGroup
cfg = load_config('kitti_models/det6d_car.yaml')
model = build_model(cfg, seed=1234, device='cuda')
b, n = 8, 16384
points = torch.from_numpy(bench.synth_points(1000, b, n)).cuda()
with torch.no_grad():
    model({'batch_size': b, 'points': points})
g = Det6DGroup(model, b, n, 1, torch.cuda.Stream(), points=points)
for _ in range(3):
    for r in g.launch():
        r.finalize()
    torch.cuda.synchronize()
