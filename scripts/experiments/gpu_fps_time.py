import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from de6d_amd.ops import pointnet2_batch_hip as pn
from tests.util import make_batch
def t_fps(b, n, m, weighted=False, reps=5):
    xyz = torch.from_numpy(np.ascontiguousarray(make_batch(1, b, n)[..., :3])).cuda()
    w = torch.rand((b, n), device='cuda')
    idx = torch.zeros((b, m), dtype=torch.int32, device='cuda')
    best = 1e9
    for _ in range(reps):
        temp = torch.full((b, n), 1e10, device='cuda')
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if weighted: pn.furthest_point_sampling_weights_wrapper(b, n, m, xyz, w, temp, idx)
        else: pn.farthest_point_sampling_wrapper(b, n, m, xyz, temp, idx)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("fps b=%d n=%d m=%d w=%d: %.3f ms  (%.3f us/round)" % (b, n, m, weighted, best * 1e3, best * 1e6 / m))
t_fps(8, 16384, 4096); t_fps(8, 4096, 512); t_fps(8, 4096, 512, True); t_fps(8, 512, 256); t_fps(8, 512, 256, True); t_fps(8, 8192, 1024)
