"""round-4 debugging aid: config 5, b = 2, fresh captured runners, the test's exact sequence; where do wrong picks come from"""
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


def where(rows, cloud):
    table = {tuple(r): i for i, r in reversed(list(enumerate(cloud.tolist())))}
    return [table.get(tuple(r), -1) for r in rows.tolist()]


for trial in range(4):
    runner = GraphedDet6D(model, b, n)
    for seed in (8100, 8200):
        pts_np = flat_points(make_batch(seed, b, n, tilt=False))
        pts = torch.from_numpy(pts_np).cuda()
        runner.launch(pts).finalize()
        torch.cuda.synchronize()
        got = runner.batch_dict['point_coords_list'][0][:, 1:].reshape(b, -1, 3).clone()
        want, widx = direct(pts)
        clouds = pts_np[:, 1:4].reshape(b, n, 3)
        for s in range(b):
            ne = (got[s] != want[s]).any(1)
            print('trial', trial, 'seed', seed, 'scene', s, 'point0', clouds[s, 0].tolist(), 'mismatching', int(ne.sum()),
                  'first', int(ne.nonzero()[0]) if ne.any() else -1, flush=True)
            if ne.any():
                g = got[s].cpu().numpy()
                for o in range(b):
                    w = where(g[:12], clouds[o])
                    print('   got rows 0..11 as indices of scene', o, ':', w)
                print('   want idx 0..11', widx[s, :12].tolist())
                first = int(ne.nonzero()[0])
                print('   around first mismatch: got', g[max(0, first - 1):first + 3].tolist(), 'want', want[s, max(0, first - 1):first + 3].tolist())
