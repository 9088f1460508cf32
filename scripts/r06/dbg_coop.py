import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from de6d_amd.ops import fused
from oracle import ops as oops
from tests.util import make_batch
oops.build()
for rep in range(3):
    for (seed, b, n, m, dup) in ((33, 2, 32768, 2048, 0.3), (35, 3, 32768, 1024, 0.0), (34, 1, 65536, 2048, 0.0)):
        xyz = make_batch(seed, b, n, dup_frac=dup)[..., :3]
        x = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
        idx = torch.full((b, m), -7, dtype=torch.int32, device='cuda')
        ws = fused.fps_workspace(b, n)
        fused.fps_fused(x, 0, n, m, None, 1.0, idx, 0, temp=ws); torch.cuda.synchronize()
        t0 = time.perf_counter()
        fused.fps_fused(x, 0, n, m, None, 1.0, idx, 0, temp=ws); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        got = idx.cpu().numpy(); ref = oops.fps(xyz, m)
        bad = np.argwhere(got != ref)
        print(rep, (b, n, m, dup), 'exact', len(bad) == 0, '%.2f ms' % (dt * 1e3), 'mismatches', len(bad), bad[:3].tolist(), [(int(got[tuple(p)]), int(ref[tuple(p)])) for p in bad[:3]], flush=True)
