"""the cooperative register-resident sampler (csrc/fps_coop.hip: 32768 / 65536-point scenes on 2 / 4 workgroups) against
the CPU oracle: bit-exact indices on scenes, lattices full of exact ties, duplicates, all-equal clouds, odd batch sizes"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from de6d_amd.ops import fused
from oracle import ops as oops
from tests.util import make_batch
oops.build()

FALLBACK = 'fallback' in sys.argv     # a plain (b, n) float scratch instead of the cooperative workspace: memory-resident kernel


def check(xyz, m, tag):
    x = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
    idx = torch.full((xyz.shape[0], m), -7, dtype=torch.int32, device='cuda')
    ws = fused.fps_workspace(xyz.shape[0], xyz.shape[1])
    if FALLBACK:
        ws = torch.empty((xyz.shape[0] * xyz.shape[1] * 4,), dtype=torch.uint8, device='cuda')
    fused.fps_fused(x, 0, xyz.shape[1], m, None, 1.0, idx, 0, temp=ws); torch.cuda.synchronize()
    t0 = time.perf_counter()
    fused.fps_fused(x, 0, xyz.shape[1], m, None, 1.0, idx, 0, temp=ws); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fused.fps_status(xyz.shape[0], xyz.shape[1], ws)
    ok = np.array_equal(idx.cpu().numpy(), oops.fps(xyz, m))
    print(tag, 'exact', ok, '%.2f ms, %.2f us/round' % (dt * 1e3, dt * 1e6 / max(m - 1, 1)), flush=True)
    return ok

allok = True
rng = np.random.default_rng(0)
if FALLBACK:      # the memory-resident fallback: two short cases
    allok &= check(make_batch(31, 2, 65536, dup_frac=0.1)[..., :3], 600, 'fallback scenes 2 x 65536')
    allok &= check(make_batch(33, 1, 32768, dup_frac=0.3)[..., :3], 400, 'fallback scenes 1 x 32768')
    print('ALL', allok)
    sys.exit(0)
allok &= check(make_batch(31, 3, 65536, dup_frac=0.1)[..., :3], 4096, 'scenes 3 x 65536')
allok &= check(make_batch(32, 9, 65536, tilt=True)[..., :3], 1024, 'scenes 9 x 65536 (two dispatch groups)')
allok &= check(make_batch(33, 2, 32768, dup_frac=0.3)[..., :3], 2048, 'scenes 2 x 32768 (two parts)')
allok &= check((rng.integers(0, 60, size=(2, 65536, 3)) * 0.25).astype(np.float32), 3000, 'lattice (many exact ties)')
allok &= check(np.ones((1, 65536, 3), np.float32) * 3.5, 700, 'all points equal')
far = rng.normal(size=(2, 65536, 3)).astype(np.float32); far[:, :7] *= 1e4
allok &= check(far, 1024, 'far outliers')
allok &= check(make_batch(34, 1, 65536)[..., :3], 16384, 'BASELINE config 5 first layer: 16384 of 65536')
print('ALL', allok)
