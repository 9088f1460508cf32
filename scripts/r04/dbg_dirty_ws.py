"""round-4 debugging aid: does the cooperative sampler depend on what its workspace held before the launch?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.util import make_batch
from de6d_amd.ops import fused
from de6d_amd import _lib as L

b, n, m = 2, 65536, 16384
cloud = torch.from_numpy(make_batch(8100, b, n, tilt=False)[..., :3].copy()).cuda().contiguous()


def run(ws):
    idx = torch.zeros(b, m, dtype=torch.int32, device='cuda')
    fused.fps_fused(cloud, 0, n, m, None, 0.0, idx, 0, temp=ws)
    torch.cuda.synchronize()
    return idx


nbytes = int(L.lib().det6d_fps_fused_workspace_bytes(b, n))
clean = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
want = run(clean)
print('workspace bytes', nbytes)
for name, fill in (('0xff', 255), ('0x7f', 127), ('0x01', 1), ('random', None)):
    ws = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device='cuda') if fill is None else torch.full((nbytes,), fill, dtype=torch.uint8, device='cuda')
    ws[:256] = 0
    got = run(ws)
    print(name, 'first run equal:', [bool(torch.equal(got[s], want[s])) for s in range(b)], 'status', int(ws[:4].view(torch.int32)))
    got = run(ws)
    print(name, 'second run equal:', [bool(torch.equal(got[s], want[s])) for s in range(b)])
# region by region
step = nbytes // 16
for i in range(16):
    ws = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
    lo, hi = max(256, i * step), (i + 1) * step
    ws[lo:hi] = torch.randint(0, 256, (hi - lo,), dtype=torch.uint8, device='cuda')
    got = run(ws)
    print('dirty sixteenth', i, [bool(torch.equal(got[s], want[s])) for s in range(b)])
