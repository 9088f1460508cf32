"""round-4 debugging aid: config 5 (65536 points), b = 2, eager vs captured graph vs the stand-alone sampler call"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.util import make_batch
from tests.test_model_gpu import flat_points
from de6d_amd.runtime import load_config, build_model, GraphedDet6D
from de6d_amd.ops import fused

cfg = load_config('synthetic_models/det6d_65536.yaml')
model = build_model(cfg, seed=77, device='cuda')
b, n = 2, 65536


def direct(pts):
    xyz = pts[:, 1:4].reshape(b, n, 3).contiguous()
    idx = torch.zeros(b, 16384, dtype=torch.int32, device='cuda')
    fused.fps_fused(xyz, 0, n, 16384, None, 0.0, idx, 0)
    torch.cuda.synchronize()
    fused.check_fps_status()
    return torch.gather(xyz, 1, idx.long()[..., None].expand(-1, -1, 3)), idx


def cmp(tag, got, want):
    got = got.reshape(b, -1, 3); want = want.reshape(b, -1, 3)
    for s in range(b):
        ne = (got[s] != want[s]).any(1)
        first = int(ne.nonzero()[0]) if ne.any() else -1
        print(tag, 'scene', s, 'mismatching picks', int(ne.sum()), 'first', first, flush=True)


runner = GraphedDet6D(model, b, n)
for rep in range(3):
    for seed in (8100, 8200):
        pts = torch.from_numpy(flat_points(make_batch(seed, b, n))).cuda()
        want, widx = direct(pts)
        want2, _ = direct(pts)
        cmp('direct-twice seed %d' % seed, want2, want)
        with torch.no_grad():
            model({'batch_size': b, 'points': pts})
            bd = {'batch_size': b, 'points': pts}
            model(bd)
        torch.cuda.synchronize()
        cmp('eager seed %d' % seed, bd['point_coords_list'][0][:, 1:], want)
        runner.launch(pts).finalize()
        torch.cuda.synchronize()
        got = runner.batch_dict['point_coords_list'][0][:, 1:]
        cmp('graph seed %d rep %d' % (seed, rep), got, want)
        if (got.reshape(b, -1, 3) != want).any():
            g = got.reshape(b, -1, 3)
            for s in range(b):
                print(' scene', s, 'got rows 0..3', g[s, :4].tolist(), 'want', want[s, :4].tolist())
