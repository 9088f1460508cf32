"""protocol counters of the look-ahead sampler (experiments build): DET6D_EXPERIMENTS_LIB=1 DET6D_FPS_SEQ=g python this [beam]"""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from de6d_amd import _lib as L
from de6d_amd.ops import fused
from tests.util import make_batch, beam_batch
names = ['seq steps', 'decisions', 'blocked', 'ring breaks', 'polls w/ new rec', 'picks replayed', 'records accepted',
         'rescans (all)', 'cyc poll/accept', 'cyc decide', 'cyc blocked try', 'cyc total', 'o4 empty polls', 'o4 rescans',
         'o4 owner steps', 'o4 extra applies']
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
print('SEQ=%s b=%d m=%d: %.3f ms, %.3f us/round' % (os.environ.get('DET6D_FPS_SEQ'), b, m, dt * 1e3, dt * 1e6 / m))
for k, v in zip(names, out):
    print('  %-18s %d' % (k, v))
