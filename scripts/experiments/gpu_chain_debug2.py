import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from de6d_amd.ops import fused
def dev(a): return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
b, n, m, ns = 1, 64, 1, 32
widths = (32, 32, 64)
rows = np.zeros((b, n, 4), np.float32)
rows[0, :, 0] = np.arange(n) * 0.01 + 1      # x: distinct per point
rows[0, :, 1] = 2; rows[0, :, 2] = 3; rows[0, :, 3] = 4
ctr = np.zeros((b, m, 3), np.float32)
idx = np.arange(32, dtype=np.int32).reshape(1, 1, 32)
cnt = np.ones((b, m), np.int32)
def run(w1, s1, w2, s2, w3, s3):
    layers = [(dev(w1), dev(s1), 32, 1), (dev(w2), dev(s2), 32, 1), (dev(w3), dev(s3), 64, 1)]
    out = torch.zeros((b * m, 64), device='cuda')
    fused.mlp_chain3(dev(rows), torch.from_numpy(idx).cuda(), dev(ctr), torch.from_numpy(cnt).cuda(), layers, out, 0)
    return out.cpu().numpy()[0]
I32 = np.eye(32, dtype=np.float32)
w3 = np.zeros((32, 64), np.float32); w3[:, :32] = I32; w3[:, 32:] = 2 * I32
# layer 1: channel c = (c+1) * y  (y = 2)  -> 2, 4, 6, ... ; pass through layers 2 and 3
w1 = np.zeros((4, 32), np.float32); w1[1, :] = np.arange(1, 33)
o = run(w1, np.zeros(32), I32, np.zeros(32), w3, np.zeros(64))
print('identity chain, expect 2,4,..,64 | 4,8,..:', o[:8], o[32:40])
print(' full', o.astype(int).tolist())
o = run(w1, np.arange(32) * 100.0, I32, np.zeros(32), w3, np.zeros(64))
print('shift1 expect 2+0, 4+100, 6+200..:', o[:6].astype(int))
o = run(w1, np.zeros(32), I32, np.arange(32) * 100.0, w3, np.zeros(64))
print('shift2 expect same:', o[:6].astype(int))
