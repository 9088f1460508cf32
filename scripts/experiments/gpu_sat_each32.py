"""each GEMM-family launch of a 32-scene pass replayed ALONE on 16 streams: its rate with the chip full, on the information
rows (NOT a result: a map of where the family's chip-full time goes)"""
import os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from de6d_amd.ops import fused
from de6d_amd.runtime import load_config, build_model
cfg = load_config('kitti_models/det6d_car.yaml')
model = build_model(cfg, seed=1234, device='cuda')
scene = sys.argv[1] if len(sys.argv) > 1 else 'uniform'
b = 32
points = torch.from_numpy(bench.synth_points(1000, b, 16384, scene=scene)).cuda()
streams = [torch.cuda.Stream() for _ in range(16)]
with torch.no_grad():
    model({'batch_size': b, 'points': points})
fused.LINEAR_REPLAY, fused.LINEAR_EVENTS = [], []
with torch.no_grad():
    model({'batch_size': b, 'points': points})
torch.cuda.synchronize()
replay, ev = fused.LINEAR_REPLAY, fused.LINEAR_EVENTS
fused.LINEAR_REPLAY = fused.LINEAR_EVENTS = None
tot_sat = tot_alone = tot_fl = 0.0
for i, (item, (e0, e1, r, k, n)) in enumerate(zip(replay, ev)):
    rows = int(r.cpu()[8]) if torch.is_tensor(r) else r
    fl = 2.0 * rows * k * n
    s = bench.family_saturated([item], reps=6, streams=streams)
    per = s['seconds'] / s['passes']
    tot_sat += per; tot_alone += e0.elapsed_time(e1) * 1e-3; tot_fl += fl
    print("%2d rows %7d K %4d N %6d  alone %7.1f us %6.1f TF | chip full %7.1f us per launch %6.1f TF" % (
        i, rows, k, n, e0.elapsed_time(e1) * 1e3, fl / e0.elapsed_time(e1) / 1e9, per * 1e6, fl / per / 1e12), flush=True)
print("%s: sum alone %.3f ms (%.1f TF), sum of per-launch chip-full times %.3f ms (%.1f TF)" % (
    scene, tot_alone * 1e3, tot_fl / tot_alone / 1e12, tot_sat * 1e3, tot_fl / tot_sat / 1e12))
s = bench.family_saturated(replay, reps=12, streams=streams)
print("whole family on 16 streams: %.3f ms per pass (%.1f TF)" % (s['seconds'] / s['passes'] * 1e3, tot_fl / (s['seconds'] / s['passes']) / 1e12))
