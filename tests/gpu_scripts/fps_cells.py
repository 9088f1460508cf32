import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from de6d_amd.ops import fused
from oracle import ops as oops
from tests.util import make_batch
def run(b, n, m, dup, check=True, seed=1):
    xyz = np.ascontiguousarray(make_batch(seed, b, n, dup_frac=dup)[..., :3])
    if dup >= 1.0: xyz[:, n // 2:] = xyz[:, :n - n // 2]
    x = torch.from_numpy(xyz).cuda()
    idx = torch.zeros((b, m), dtype=torch.int32, device='cuda')
    fused.fps_fused(x, 0, n, m, None, 1.0, idx, 0); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fused.fps_fused(x, 0, n, m, None, 1.0, idx, 0)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    ok = None
    if check:
        ref = oops.fps(xyz[:2], m)
        ok = np.array_equal(idx.cpu().numpy()[:2], ref)
        if not ok:
            g = idx.cpu().numpy()[:2]; bad = np.argwhere(g != ref)
            print('   first mismatch at', bad[0], g[tuple(bad[0])], ref[tuple(bad[0])])
    print("n=%d m=%d dup=%.2f: %.3f ms (%.3f us/round) exact=%s" % (n, m, dup, best * 1e3, best * 1e6 / m, ok))
run(8, 16384, 4096, 0.05); run(8, 16384, 4096, 1.0); run(2, 16384, 300, 0.0, seed=5)
run(8, 8192, 1024, 0.1); run(8, 4096, 512, 0.1); run(8, 4096, 512, 1.0)
x = np.ones((2, 16384, 3), np.float32); idx = torch.zeros((2, 64), dtype=torch.int32, device='cuda')
fused.fps_fused(torch.from_numpy(x).cuda(), 0, 16384, 64, None, 1.0, idx, 0)
print('all-equal exact', np.array_equal(idx.cpu().numpy(), oops.fps(x, 64)))
