"""extra seeds / cloud shapes for the wave-skip sampler against the CPU oracle (bit-exact indices)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from de6d_amd.ops import fused
from oracle import ops as oops
from tests.util import make_batch
def check(xyz, m, tag):
    x = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
    idx = torch.zeros((xyz.shape[0], m), dtype=torch.int32, device='cuda')
    fused.fps_fused(x, 0, xyz.shape[1], m, None, 1.0, idx, 0); torch.cuda.synchronize()
    ok = np.array_equal(idx.cpu().numpy(), oops.fps(xyz, m))
    print(tag, 'exact', ok, flush=True)
    return ok
allok = True
for seed in range(20, 26):
    allok &= check(make_batch(seed, 4, 16384, tilt=seed % 2 == 0, dup_frac=0.1 * (seed % 3))[..., :3], 4096, 'scene seed %d' % seed)
rng = np.random.default_rng(0)
allok &= check(rng.normal(size=(2, 16384, 3)).astype(np.float32), 4096, 'gaussian blob')
allok &= check((rng.integers(0, 40, size=(2, 16384, 3)) * 0.25).astype(np.float32), 4096, 'lattice (many exact ties)')
line = np.zeros((2, 16384, 3), np.float32); line[..., 0] = rng.uniform(0, 70, (2, 16384))
allok &= check(line, 2048, 'points on a line')
far = rng.normal(size=(2, 16384, 3)).astype(np.float32); far[:, :7] *= 1e4
allok &= check(far, 1024, 'far outliers')
print('ALL', allok)
