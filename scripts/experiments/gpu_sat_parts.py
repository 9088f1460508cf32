"""GEMM family with the chip full, split: chain launches only / linear launches only / the few-row FC launches only (NOT a result)"""
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
streams = [torch.cuda.Stream() for _ in range(16)]
with torch.no_grad():
    model({'batch_size': 8, 'points': points})
fused.LINEAR_REPLAY, fused.LINEAR_EVENTS = [], []
with torch.no_grad():
    model({'batch_size': 8, 'points': points})
torch.cuda.synchronize()
replay, ev = fused.LINEAR_REPLAY, fused.LINEAR_EVENTS
fused.LINEAR_REPLAY = fused.LINEAR_EVENTS = None
def flops(i):
    _, _, r, k, n = ev[i]
    rows = int(r.cpu()[0]) if torch.is_tensor(r) else r
    return 2.0 * rows * k * n
groups = {"all": list(range(len(replay))),
          "chains (K == 1 entries)": [i for i in range(len(replay)) if ev[i][3] == 1],
          "linear on compact lists": [i for i in range(len(replay)) if ev[i][3] != 1 and torch.is_tensor(ev[i][2])],
          "plain FC layers": [i for i in range(len(replay)) if ev[i][3] != 1 and not torch.is_tensor(ev[i][2])]}
for name, ids in groups.items():
    s = bench.family_saturated([replay[i] for i in ids], reps=16, streams=streams)
    per = s['seconds'] / s['passes']
    fl = sum(flops(i) for i in ids)
    print("%-26s %2d launches  %.3f ms per pass  %6.2f GF  %6.1f TF" % (name, len(ids), per * 1e3, fl / 1e9, fl / per / 1e12), flush=True)
