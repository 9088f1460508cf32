"""round-4 debugging aid: the cooperative sampler alone inside a captured graph, workspace holding junk"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.util import make_batch
from de6d_amd.ops import fused
from de6d_amd import _lib as L

b, n, m = 2, 65536, 16384
clouds = {s: torch.from_numpy(make_batch(s, b, n, tilt=False)[..., :3].copy()).cuda().contiguous() for s in (8100, 8200)}
nbytes = int(L.lib().det6d_fps_fused_workspace_bytes(b, n))
A = lambda v: (v + 255) & ~255
items = b * n * 4
off = {'err': 0, 'keys_in': 256}
off['keys_out'] = A(off['keys_in'] + items); off['vals_in'] = A(off['keys_out'] + items); off['vals_out'] = A(off['vals_in'] + items)
off['cub'] = A(off['vals_out'] + items)
off['exch'] = nbytes - A(b * 2592 * 8)
print('bytes', nbytes, off)


def eager(cloud):
    ws = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
    idx = torch.zeros(b, m, dtype=torch.int32, device='cuda')
    fused.fps_fused(cloud, 0, n, m, None, 0.0, idx, 0, temp=ws)
    torch.cuda.synchronize()
    return idx


want = {s: eager(c) for s, c in clouds.items()}
stream = torch.cuda.Stream()


def graphed(fill):
    """fill(ws) prepares the workspace content before the first replay"""
    ws = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
    idx = torch.zeros(b, m, dtype=torch.int32, device='cuda')
    src = torch.zeros_like(clouds[8100])
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=stream):
        fused.fps_fused(src, 0, n, m, None, 0.0, idx, 0, temp=ws)
    torch.cuda.synchronize()
    fill(ws)
    ws[:256] = 0
    torch.cuda.synchronize()
    res = []
    for s in (8100, 8200, 8100):
        src.copy_(clouds[s])
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        res.append([bool(torch.equal(idx[i], want[s][i])) for i in range(b)])
    return res


def junk(lo, hi):
    def f(ws):
        ws[lo:hi] = torch.randint(0, 256, (hi - lo,), dtype=torch.uint8, device='cuda')
    return f


def previous_run(ws):
    idx = torch.zeros(b, m, dtype=torch.int32, device='cuda')
    fused.fps_fused(clouds[8200], 0, n, m, None, 0.0, idx, 0, temp=ws)
    torch.cuda.synchronize()


print('zero workspace        ', graphed(lambda ws: None))
print('previous run leftovers', graphed(previous_run))
print('junk everywhere       ', graphed(junk(256, nbytes)))
names = list(off)
for i, name in enumerate(names[1:], 1):
    hi = off[names[i + 1]] if i + 1 < len(names) else nbytes
    print('junk in %-9s' % name, graphed(junk(off[name], hi)))
print('junk 0xff in exch     ', graphed(lambda ws: ws[off['exch']:].fill_(255)))
