"""first-layer sampler on 32 scenes per launch (what a group's stage 1 issues), benchmark and ray-cast scenes: ms per launch"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from de6d_amd.ops import fused
from tests.util import make_batch, beam_batch
n, m, b = 16384, 4096, 32
for name, xyz in (('uniform', make_batch(1000, b, n)[..., :3]), ('ray-cast', beam_batch(1000, b, n)[..., :3])):
    x = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
    idx = torch.zeros((b, m), dtype=torch.int32, device='cuda')
    ws = fused.fps_workspace(b, n)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fused.fps_fused(x, 0, n, m, None, 1.0, idx, 0, temp=ws)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    per = []
    for s in range(0, b, 4):      # which scenes are the slow ones
        xs = x[s:s + 1].contiguous(); i1 = torch.zeros((1, m), dtype=torch.int32, device='cuda'); w1 = fused.fps_workspace(1, n)
        fused.fps_fused(xs, 0, n, m, None, 1.0, i1, 0, temp=w1); torch.cuda.synchronize(); t0 = time.perf_counter()
        fused.fps_fused(xs, 0, n, m, None, 1.0, i1, 0, temp=w1); torch.cuda.synchronize(); per.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(name, 'b=32: min %.3f ms' % min(ts[1:]), 'single scenes:', per, flush=True)
