"""round 6 (review item 2): every GEMM-family launch of an 80-scene pass replayed ALONE in a loop for ~1 s on one stream:
time per launch, TFLOP/s on the information rows, shader clock and socket power while nothing else runs.  Is the clock pulled
down under the wide group's weight stream (the review's hypothesis), and what does each kernel cost in Joules per TFLOP?"""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
from de6d_amd.runtime import load_config, build_model  # noqa: E402
from de6d_amd.ops import fused  # noqa: E402
from de6d_amd import synthetic  # noqa: E402
from bench_legs import ClockPowerSampler  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else 'uniform'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
only = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else None      # launch numbers (issue order) to replay
cfg = load_config('kitti_models/det6d_car.yaml')
model = build_model(cfg, seed=1234, device='cuda')
b, n = 80, 16384
make = synthetic.beam_batch if scene == 'beam' else synthetic.make_batch
pts = torch.from_numpy(synthetic.points_tensor(make(1000, b, n))).cuda()
with torch.no_grad():
    model({'batch_size': b, 'points': pts})
    fused.LINEAR_EVENTS, fused.LINEAR_REPLAY = [], []
    model({'batch_size': b, 'points': pts})
torch.cuda.synchronize()
ev, replay = fused.LINEAR_EVENTS, fused.LINEAR_REPLAY
fused.LINEAR_EVENTS = fused.LINEAR_REPLAY = None
assert len(ev) == len(replay)
print("%-4s %-10s %-8s %9s %8s %8s %8s %8s" % ("#", "rows", "flop/row", "us", "TFLOP/s", "sclk", "W", "J/TFLOP"))
ident = lambda t: t.data_ptr()
for i, ((e0, e1, r, k, nn), entry) in enumerate(zip(ev, replay)):
    if only is not None and i not in only:
        continue
    rows = r.cpu().tolist()[8] if torch.is_tensor(r) else r
    issue = entry[0]
    us0 = e0.elapsed_time(e1) * 1e3
    reps = max(20, int(secs * 1e6 / max(us0, 5.0)))
    for _ in range(5):
        issue(ident)
    torch.cuda.synchronize()
    smp = ClockPowerSampler(0.01)
    time.sleep(0.15)
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    a.record()
    for _ in range(reps):
        issue(ident)
    z.record()
    torch.cuda.synchronize()
    t1 = time.time()
    c = smp.stop(t0 + 0.25 * (t1 - t0), t1) or {}
    us = a.elapsed_time(z) * 1e3 / reps
    tf = 2.0 * rows * k * nn / us / 1e6
    pw = c.get('power_w')
    print("%-4d %-10d %-8d %9.1f %8.1f %8s %8s %8s" % (i, rows, 2 * k * nn, us, tf, c.get('sclk_mhz'), pw,
                                                     ('%.1f' % (pw / tf)) if pw else None), flush=True)
    time.sleep(0.3)
