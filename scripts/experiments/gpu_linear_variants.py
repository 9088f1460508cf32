import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from de6d_amd.ops import fused
shapes = [(65536, 512, 1024), (65536, 256, 512), (131072, 256, 256), (131072, 128, 256), (262144, 96, 128), (65536, 128, 128)]
for r, k, n in shapes:
    a = torch.randn((r, k), device='cuda'); w = torch.randn((k, n), device='cuda') / k ** 0.5
    sh = torch.randn((n,), device='cuda'); out = torch.empty((r, n), device='cuda')
    for _ in range(3): fused.linear(a, w, sh, 1, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fused.linear(a, w, sh, 1, out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("rows %7d K %4d N %4d  %8.1f us  %6.1f TF" % (r, k, n, dt * 1e6, 2.0 * r * k * n / dt / 1e12))
