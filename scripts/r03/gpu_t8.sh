cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
python3 -m pytest tests/test_ops_gpu.py tests/test_fp_gpu.py -x -q -m gpu -k "three or fp or propagation" 2>&1 | tail -3
python3 - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from de6d_amd.ops import pointnet2_batch_hip as pn2
b, c, m, n = 8, 64, 4096, 16384
feats = torch.randn((b, c, m), device='cuda'); i3 = torch.randint(0, m, (b, n, 3), dtype=torch.int32, device='cuda')
w3 = torch.rand((b, n, 3), device='cuda'); out = torch.empty((b, c, n), device='cuda')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for rep in range(3):
    ev[0].record()
    for _ in range(10): pn2.three_interpolate_wrapper(b, c, m, n, feats, i3, w3, out)
    ev[1].record(); torch.cuda.synchronize()
t = ev[0].elapsed_time(ev[1]) / 10 * 1e-3
print("three_interpolate (8,64,4096)->16384: %.1f us, %.0f GB/s algorithmic" % (t * 1e6, b * (c * m * 4 + n * 24 + c * n * 4) / t / 1e9))
PY
