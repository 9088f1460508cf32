"""round-4 debugging aid: after a wrong first replay, look into the sampler's workspace"""
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
A = lambda v: (v + 255) & ~255


def direct(pts):
    xyz = pts[:, 1:4].reshape(b, n, 3).contiguous()
    idx = torch.zeros(b, 16384, dtype=torch.int32, device='cuda')
    fused.fps_fused(xyz, 0, n, 16384, None, 0.0, idx, 0)
    torch.cuda.synchronize()
    return torch.gather(xyz, 1, idx.long()[..., None].expand(-1, -1, 3)), idx


for trial in range(4):
    runner = GraphedDet6D(model, b, n)
    for seed in (8100, 8200):
        pts_np = flat_points(make_batch(seed, b, n, tilt=False))
        pts = torch.from_numpy(pts_np).cuda()
        runner.launch(pts).finalize()
        torch.cuda.synchronize()
        got = runner.batch_dict['point_coords_list'][0][:, 1:].reshape(b, -1, 3).clone()
        word = runner._status_words[0]
        st = word.untyped_storage()
        raw = torch.empty(0, dtype=torch.uint8, device='cuda').set_(st, word.storage_offset() * 4, (st.nbytes() - word.storage_offset() * 4,))
        want, widx = direct(pts)
        bad = [bool((got[s] != want[s]).any()) for s in range(b)]
        print('trial', trial, 'seed', seed, 'bad', bad, 'n status words', len(runner._status_words), 'ws bytes', raw.numel(), flush=True)
        items = b * n * 4
        off_keys_in = 256
        off_keys_out = A(off_keys_in + items)
        off_vals_in = A(off_keys_out + items)
        off_vals_out = A(off_vals_in + items)
        perm = raw[off_vals_out:off_vals_out + items].view(torch.int32).reshape(b, n).cpu().numpy()
        keys = raw[off_keys_out:off_keys_out + items].view(torch.int32).reshape(b, n).cpu().numpy()
        for s in range(b):
            ok = np.array_equal(np.sort(perm[s]), np.arange(n))
            print('   scene', s, 'perm is a permutation:', ok, 'keys sorted:', bool((np.diff(keys[s].astype(np.int64)) >= 0).all()),
                  'perm min/max', perm[s].min(), perm[s].max())
            if not ok:
                badpos = np.nonzero((perm[s] < 0) | (perm[s] >= n))[0]
                print('   out-of-range entries', len(badpos), 'first positions', badpos[:10], 'values', perm[s][badpos[:10]])
