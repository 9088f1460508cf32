"""round 6 (experiments build): phase timers of ONE mlp_group_kernel instantiation (DET6D_GROUP_PHASE_C2 / _C3 select it; wall
clock of wave 0 summed over workgroups) over back-to-back replays of that launch of an 80-scene pass"""
import ctypes, os, sys
os.environ['DET6D_EXPERIMENTS_LIB'] = '1'
os.environ.setdefault('DET6D_GROUP_STREAM', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from de6d_amd.ops import fused as F
from de6d_amd.runtime import load_config, build_model
from de6d_amd import synthetic

which = int(sys.argv[1])          # launch number in issue order (8: SA3 wide, 7: SA3 narrow, 13: head wide)
scene = sys.argv[2] if len(sys.argv) > 2 else 'uniform'
cfg = load_config('kitti_models/det6d_car.yaml')
model = build_model(cfg, seed=1234, device='cuda')
b, n = 80, 16384
make = synthetic.beam_batch if scene == 'beam' else synthetic.make_batch
pts = torch.from_numpy(synthetic.points_tensor(make(1000, b, n))).cuda()
with torch.no_grad():
    model({'batch_size': b, 'points': pts})
    F.LINEAR_EVENTS, F.LINEAR_REPLAY = [], []
    model({'batch_size': b, 'points': pts})
torch.cuda.synchronize()
replay = F.LINEAR_REPLAY
F.LINEAR_EVENTS = F.LINEAR_REPLAY = None
issue = replay[which][0]
ident = lambda t: t.data_ptr()
buf = (ctypes.c_ulonglong * 26)()
for _ in range(5):
    issue(ident)
torch.cuda.synchronize()
F.L.lib().det6d_dbg_group_phase(buf)
reps = 50
a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    issue(ident)
z.record()
torch.cuda.synchronize()
F.L.lib().det6d_dbg_group_phase(buf)
v = list(buf)
wgs, tiles = max(v[7], 1), max(v[5], 1)
names = ['layer1+barrier', 'layer2 K loop', 'layer2 epilogue+barrier', 'layer3 K loop', 'pool+store']
tot = sum(v[:5])
print('launch', which, scene, 'us per launch %.1f' % (a.elapsed_time(z) * 1e3 / reps), 'workgroups', wgs // reps, 'tiles', tiles // reps,
      'kernel us per workgroup', round(v[6] / wgs / 100.0, 1), 'us per tile', round(tot / tiles / 100.0, 2))
print('   shader clock during the kernel: %.0f MHz' % (v[8] / max(v[6], 1) * 100.0))
print('   layer 3 K loop by wave (us per tile):', [round(x / tiles / 100.0, 1) for x in v[9:17]])
print('   layer 2 K loop by wave (us per tile):', [round(x / tiles / 100.0, 1) for x in v[17:25]])
for nm, x in zip(names, v[:5]):
    print('   %-26s %7.2f us per tile  %5.1f %%' % (nm, x / tiles / 100.0, 100.0 * x / tot))
