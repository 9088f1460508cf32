"""each GEMM-family launch of a pass replayed ALONE on 16 streams: its rate with the chip full (NOT a result)"""
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
fused.LINEAR_REPLAY, fused.LINEAR_EVENTS = [], []
with torch.no_grad():
    model({'batch_size': 8, 'points': points})
torch.cuda.synchronize()
replay, ev = fused.LINEAR_REPLAY, fused.LINEAR_EVENTS
fused.LINEAR_REPLAY = fused.LINEAR_EVENTS = None
tot_sat = 0.0
for i, (item, (e0, e1, r, k, n)) in enumerate(zip(replay, ev)):
    rows = int(r.cpu()[0]) if torch.is_tensor(r) else r
    fl = 2.0 * rows * k * n
    s = bench.family_saturated([item], reps=6)
    per = s['seconds'] / s['passes']
    tot_sat += per
    print("%2d rows %7d K %4d N %5d  alone %7.1f us %6.1f TF | chip full %7.1f us per launch %6.1f TF" % (
        i, rows, k, n, e0.elapsed_time(e1) * 1e3, fl / e0.elapsed_time(e1) / 1e9, per * 1e6, fl / per / 1e12), flush=True)
print("sum of per-launch chip-full times: %.3f ms" % (tot_sat * 1e3))
for i in (16, 9, 5, 25, 7):
    for ns in (1, 2, 4, 8, 16):
        s = bench.family_saturated([replay[i]], n_streams=ns, reps=6)
        print("launch %2d on %2d streams: %.1f us per launch" % (i, ns, s['seconds'] / s['passes'] * 1e6), flush=True)
