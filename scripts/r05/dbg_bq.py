"""debug: which centres of the grid query differ from the oracle, and how"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util import beam_batch
from de6d_amd.ops import fused
from oracle import ops as oops
oops.build()
n, ns = 16384, 16
full = beam_batch(11, 2, n)[..., :3]
for label, xyz, m in (("both scenes m=96", full, 96), ("scene 1 alone m=96", full[1:2], 96), ("scene 0 alone m=96", full[0:1], 96), ("swapped m=96", full[::-1], 96),
                      ("both m=256", full, 256), ("both m=64", full, 64), ("both m=200", full, 200)):
    xyz = np.ascontiguousarray(xyz)
    b = xyz.shape[0]
    new_xyz = np.ascontiguousarray(xyz[:, ::n // m][:, :m] + np.float32(0.01))
    ocnt, oidx = oops.ball_query_dilated(0.4, 2.5, ns, xyz, new_xyz)
    ca, ia, cb, ib = fused.ball_query_pair(torch.from_numpy(xyz).cuda(), torch.from_numpy(new_xyz).cuda(), (0.4, 2.5, ns), (0.0, 0.0, 1), grid=True)
    ca, ia = ca.cpu().numpy(), ia.cpu().numpy()
    bad = (ia != oidx).any(-1)
    print(label, "cnt equal", (ca == ocnt).all(), "bad centres", int(bad.sum()), "of", b * m)
    for bi in range(b):
        ids = np.nonzero(bad[bi])[0]
        print("   scene", bi, "bad:", (int(ids.min()), int(ids.max()), len(ids)) if len(ids) else None,
              "zero rows:", int((ia[bi][bad[bi]] == 0).all(-1).sum()) if len(ids) else 0)
