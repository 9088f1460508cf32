"""GEMM family replayed on 16 streams (bench.family_saturated) for a kernel trace (NOT a result)"""
import os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from de6d_amd.ops import fused
from de6d_amd.runtime import load_config, build_model
cfg = load_config('kitti_models/det6d_car.yaml')
model = build_model(cfg, seed=1234, device='cuda')
points = torch.from_numpy(bench.synth_points(1000, 8, 16384)).cuda()
with torch.no_grad():
    model({'batch_size': 8, 'points': points})
fused.LINEAR_REPLAY = []
with torch.no_grad():
    model({'batch_size': 8, 'points': points})
torch.cuda.synchronize()
replay = fused.LINEAR_REPLAY
fused.LINEAR_REPLAY = None
r = bench.family_saturated(replay, reps=4)
print("family ms per pass", r['seconds'] / r['passes'] * 1e3)
