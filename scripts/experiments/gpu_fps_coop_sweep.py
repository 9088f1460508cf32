"""cooperative multi-pick sampler, 16384 of 65536 points (BASELINE config 5's first layer): ms per launch of b scenes;
run under DET6D_EXPERIMENTS_LIB=1 with DET6D_FPS_SEQ_PICKS / DET6D_FPS_COOP_MULTI / DET6D_FPS_COOP_FAST"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from de6d_amd.ops import fused
from tests.util import make_batch, beam_batch
b = int(os.environ.get('B', 8))
n, m = 65536, 16384
xyz = (beam_batch(34, b, n) if 'beam' in sys.argv else make_batch(34, b, n))[..., :3]
x = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
idx = torch.zeros((b, m), dtype=torch.int32, device='cuda')
ws = fused.fps_workspace(b, n)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fused.fps_fused(x, 0, n, m, None, 1.0, idx, 0, temp=ws)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
fused.fps_status(b, n, ws)
print('b=%d picks cap %s multi %s fast %s: %.2f ms, %.3f us/pick' % (b, os.environ.get('DET6D_FPS_SEQ_PICKS'), os.environ.get('DET6D_FPS_COOP_MULTI'),
      os.environ.get('DET6D_FPS_COOP_FAST'), best * 1e3, best * 1e6 / m), 'checksum', int(idx.sum()))
