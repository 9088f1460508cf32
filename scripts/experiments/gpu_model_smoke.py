import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from de6d_amd.runtime import load_config, build_model
from tests.util import make_batch
cfg = load_config('kitti_models/det6d_car.yaml')
model = build_model(cfg, device='cuda')
B, N = 8, 16384
pts = make_batch(1000, B, N)
flat = np.concatenate([np.repeat(np.arange(B, dtype=np.float32), N)[:, None], pts.reshape(B * N, 4)], 1)
points = torch.from_numpy(flat).cuda()
with torch.no_grad():
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        pred, _ = model({'batch_size': B, 'points': points})
        torch.cuda.synchronize(); print('forward %.2f ms' % ((time.time() - t0) * 1e3), [len(p['pred_scores']) for p in pred])
    # per-stage timing
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        pred, _ = model({'batch_size': B, 'points': points}); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25))
print(pred[0]['pred_boxes'][:3])
