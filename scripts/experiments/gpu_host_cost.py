"""host-side cost of launching one pass (NOT a result): time spent inside the launch calls, GPU work stubbed out"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from de6d_amd.ops import fused
from de6d_amd.runtime import load_config, build_model, Det6DGroup, GraphedDet6D
cfg = load_config('kitti_models/det6d_car.yaml')
model = build_model(cfg, seed=1234, device='cuda')
b, n, k = 8, 16384, 4
points = torch.from_numpy(bench.synth_points(1000, b, n)).cuda()
with torch.no_grad():
    model({'batch_size': b, 'points': points})
mains = [torch.cuda.Stream() for _ in range(4)]
groups = [Det6DGroup(model, b, n, k, torch.cuda.Stream(), points=points, main_streams=mains) for _ in range(2)]
single = GraphedDet6D(model, b, n, points=points)
torch.cuda.synchronize()
for name, fn, per in (("group launch_front+launch_rest (4 passes)", lambda: [groups[i % 2].launch_front().launch_rest() for i in range(10)], 40),
                      ("single-graph pass", lambda: [single.launch() for _ in range(20)], 20)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-45s host %.3f ms per pass; with GPU drain %.3f ms per pass" % (name, (t1 - t0) / per * 1e3, (t2 - t0) / per * 1e3))
nodes = [g.segments for g in groups[0].runners][0]
print("segments per pass:", len(nodes))
