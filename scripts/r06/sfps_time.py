"""round 6: score-weighted sampler through det6d_fps_fused: exactness against the oracle + us per pick
(DET6D_KNOBS_LIB=1 DET6D_FPS_SEQW=0 = the one-pick fat-thread kernel)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from de6d_amd.ops import fused
from oracle import ops as oops
from tests.util import make_batch, beam_batch
oops.build()
ok_all = True
for (name, n, m, b, maker) in (('uniform', 4096, 512, 8, make_batch), ('ray-cast', 4096, 512, 8, beam_batch), ('uniform', 16384, 2048, 8, make_batch),
                               ('ray-cast', 16384, 2048, 4, beam_batch), ('uniform m=n', 4096, 4096, 1, make_batch)):
    pts = maker(300 + n, b, n)
    xyz = np.ascontiguousarray(pts[..., :3])
    rng = np.random.default_rng(n + m)
    scores = (rng.normal(size=(b, n)) * 2).astype(np.float32)
    x, sc = torch.from_numpy(xyz).cuda(), torch.from_numpy(scores).cuda()
    idx = torch.zeros((b, m), dtype=torch.int32, device='cuda')
    ws = fused.fps_workspace(b, n)
    fused.fps_fused(x, 0, n, m, sc, 1.0, idx, 0, temp=ws); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fused.fps_fused(x, 0, n, m, sc, 1.0, idx, 0, temp=ws)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    want = np.zeros((min(b, 2), m), np.int32)
    oops.fps_fused(xyz[:2], 0, n, m, scores[:2], 1.0, want, 0)
    got = idx.cpu().numpy()[:2]
    ok = np.array_equal(got, want); ok_all &= ok
    bad = np.argwhere(got != want)
    print('%-12s b=%d n=%d m=%d: %.3f ms (%.3f us/pick) exact=%s %s' % (name, b, n, m, best * 1e3, best * 1e6 / m, ok, bad[:2].tolist()), flush=True)
print('ALL', ok_all)
