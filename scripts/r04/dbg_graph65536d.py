"""round-4 debugging aid: which ingredient makes the first replay of a second captured runner wrong (MODE env)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.util import make_batch
from tests.test_model_gpu import flat_points
from de6d_amd.runtime import load_config, build_model, GraphedDet6D
from de6d_amd.ops import fused
from de6d_amd import _lib as L

mode = os.environ.get('MODE', 'plain')
from de6d_amd import runtime as _rt
hoists = []
_orig_init = _rt._InlineHoist.__init__


def _init(self, *a, **k):
    _orig_init(self, *a, **k)
    hoists.append(self)


_rt._InlineHoist.__init__ = _init
cfg = load_config('synthetic_models/det6d_65536.yaml')
model = build_model(cfg, seed=77, device='cuda')
b, n = 2, 65536
A = lambda v: (v + 255) & ~255

if mode == 'zeros':
    def zeros_ws(b_, n_, device='cuda'):
        return torch.zeros((int(L.lib().det6d_fps_fused_workspace_bytes(b_, n_)),), dtype=torch.uint8, device=device)
    fused.fps_workspace = zeros_ws
if mode == 'dirtyall':
    junk = [torch.randint(0, 2 ** 31 - 1, (64 << 20,), dtype=torch.int32, device='cuda') for _ in range(8)]
    del junk

pts_all = {seed: flat_points(make_batch(seed, b, n, tilt=False)) for seed in (8100, 8200)}
xyz0 = torch.from_numpy(pts_all[8100]).cuda()[:, 1:4].reshape(b, n, 3).contiguous()
ref = {}
for trial in range(3):
    runner = GraphedDet6D(model, b, n, warmup=0 if mode == 'nowarm' else 2)
    if runner._status_words:
        word = runner._status_words[0]
        st = word.untyped_storage()
        raw0 = torch.empty(0, dtype=torch.uint8, device='cuda').set_(st, word.storage_offset() * 4, (st.nbytes() - word.storage_offset() * 4,))
        nb = int(L.lib().det6d_fps_fused_workspace_bytes(b, n))
        exch0 = raw0[nb - A(b * 2592 * 8):nb].view(torch.int64).reshape(b, 2592).cpu().numpy()
        tags0 = (exch0 >> 32) & 0xffffffff
        print(mode, 'trial', trial, 'before the first replay: storage bytes', st.nbytes(), 'workspace bytes', nb, 'nonzero bytes', int((raw0[:nb] != 0).sum()),
              'exch tags min/max per scene', [(int(tags0[s_].min()), int(tags0[s_].max())) for s_ in range(b)], flush=True)
        if mode == 'zerocoop':
            raw0.zero_()
        if mode.startswith('zeropart'):
            k = int(mode[8:])
            step = nb // 8
            raw0[k * step:(k + 1) * step].zero_()
    if mode == 'zerohoist':
        for w in hoists[-1].ws.values():
            w.zero_()
        for w in hoists[-1].idx.values():
            w.zero_()
        for w in hoists[-1].ctr.values():
            w.zero_()
    torch.cuda.synchronize()
    for seed in (8100, 8200):
        pts = torch.from_numpy(pts_all[seed]).cuda()
        runner.launch(pts).finalize()
        torch.cuda.synchronize()
        got = runner.batch_dict['point_coords_list'][0][:, 1:].reshape(b, -1, 3).clone()
        if (trial, seed) == (0, 8100) or seed not in ref:
            ref[seed] = got
        bad = [bool((got[s] != ref[seed][s]).any()) for s in range(b)]
        print(mode, 'trial', trial, 'seed', seed, 'differs from trial 0:', bad, flush=True)
        if any(bad) and runner._status_words:
            word = runner._status_words[0]
            st = word.untyped_storage()
            raw = torch.empty(0, dtype=torch.uint8, device='cuda').set_(st, word.storage_offset() * 4, (st.nbytes() - word.storage_offset() * 4,))
            words = 2592
            exch_off = raw.numel() - A(b * words * 8)
            ex = raw[exch_off:exch_off + b * words * 8].view(torch.int64).reshape(b, words).cpu().numpy()
            for s in range(b):
                if not bad[s]:
                    continue
                sl = ex[s, 32:].reshape(2, 4, 64, 5)
                tags = (sl >> 32) & 0xffffffff
                pay = sl & 0xffffffff
                print('  scene', s, 'tags per parity/part (min, max):', [[(int(tags[p, q].min()), int(tags[p, q].max())) for q in range(4)] for p in range(2)])
                for p in range(2):
                    for q in range(4):
                        for c in range(64):
                            k = int(pay[p, q, c, 1]) & 0xffff
                            if k == 65024 or c < 2:
                                v = np.array([pay[p, q, c, 0], pay[p, q, c, 2], pay[p, q, c, 3], pay[p, q, c, 4]], dtype=np.uint32).view(np.float32)
                                print('   parity', p, 'part', q, 'cand', c, 'k', k, 'nc', (int(pay[p, q, c, 1]) >> 16) & 7, 'v,x,y,z', v.tolist(),
                                      'true xyz', xyz0[s, k].tolist() if seed == 8100 else None)
