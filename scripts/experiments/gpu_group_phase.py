"""timing experiment (experiments build): where a workgroup of mlp_group_kernel<256, 512, 1024> spends its time, by phase
(wall clock of wave 0, summed over workgroups), on one eager 32-scene pass (uniform and ray-cast scenes)"""
import ctypes, os, sys
os.environ['DET6D_EXPERIMENTS_LIB'] = '1'
os.environ['DET6D_GROUP_STREAM'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import synth_points
from de6d_amd.ops import fused as F
from de6d_amd.runtime import load_config, build_model

cfg = load_config('kitti_models/det6d_car.yaml')
model = build_model(cfg, seed=1234, device='cuda')
buf = (ctypes.c_ulonglong * 26)()
for scene in ('uniform', 'beam'):
    b, n = 32, 16384
    pts = torch.from_numpy(synth_points(1000, b, n, scene=scene)).cuda()
    with torch.no_grad():
        for rep in range(3):
            model({'batch_size': b, 'points': pts})
            torch.cuda.synchronize()
            F.L.lib().det6d_dbg_group_phase(buf)
    v = list(buf)
    wgs, tiles = max(v[7], 1), max(v[5], 1)
    names = ['layer1+barrier', 'layer2 K loop', 'layer2 epilogue+barrier', 'layer3 K loop', 'pool+store']
    tot = sum(v[:5])
    print(scene, 'workgroups', wgs, 'tiles', tiles, 'kernel us per workgroup', round(v[6] / wgs / 100.0, 1), 'us per tile', round(tot / tiles / 100.0, 2))
    print('   shader clock during the kernel: %.0f MHz' % (v[8] / max(v[6], 1) * 100.0))
    print('   layer 3 K loop by wave (us per tile):', [round(x / tiles / 100.0, 1) for x in v[9:17]])
    print('   layer 2 K loop by wave (us per tile):', [round(x / tiles / 100.0, 1) for x in v[17:25]])
    for nm, x in zip(names, v[:5]):
        print('   %-26s %7.2f us per tile  %5.1f %%' % (nm, x / tiles / 100.0, 100.0 * x / tot))
