"""counters of the multi-pick sampler (experiments build): DET6D_EXPERIMENTS_LIB=1 [DET6D_FPS_SEQ_PICKS=j] python this [beam]"""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from de6d_amd import _lib as L
from de6d_amd.ops import fused
from tests.util import make_batch, beam_batch
names = ['rounds', 'picks', 'rescans', 'applies', 'ended unknown', 'cyc total', 'cycA w0', 'cycA w5', 'cyc wait1', 'cycB']
n, m = 16384, int(os.environ.get('M', 4096))
b = int(os.environ.get('B', 1))
xyz = (beam_batch(3, b, n) if 'beam' in sys.argv else make_batch(1, b, n, dup_frac=0.05))[..., :3]
x = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
idx = torch.zeros((b, m), dtype=torch.int32, device='cuda')
temp = fused.fps_workspace(b, n)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fused.fps_fused(x, 0, n, m, None, 1.0, idx, 0, temp=temp)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
out = (ctypes.c_ulonglong * 16)()
fn = L.lib().det6d_dbg_fps_seq_stats
fn.argtypes = [ctypes.c_void_p]
fn(out)
print('picks/round cap %s b=%d m=%d: %.3f ms, %.3f us/pick' % (os.environ.get('DET6D_FPS_SEQ_PICKS'), b, m, dt * 1e3, dt * 1e6 / m),
      ' '.join('%s=%d' % (k, v) for k, v in zip(names, out)))
