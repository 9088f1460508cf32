"""round 6: det6d_linear stand-alone on the four plain GEMM shapes of an 80-scene pass (per-point first-layer sums of SA3 and of
the head, SA3's aggregation FC, the head's shared FC), HIP events over 20 launches each on an idle chip; prints us, TFLOP/s and a
checksum of the outputs (variants must agree bit for bit).  Environment knobs are read by the experiments library."""
import hashlib
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from de6d_amd.ops import fused  # noqa: E402

SHAPES = [(81920, 132, 256, "SA3 P sums"), (40960, 512, 256, "SA3 aggregation"), (40960, 260, 512, "head P sums"),
          (20480, 1536, 512, "head shared FC")]
torch.manual_seed(0)
tot = 0.0
tag = os.environ.get('TAG', '')
for rows, k, n, name in SHAPES:
    a = torch.randn((rows, k), device='cuda')
    w = torch.randn((k, n), device='cuda') * 0.05
    sh = torch.randn((n,), device='cuda')
    out = torch.zeros((rows, n), device='cuda')
    for _ in range(3):
        fused.linear(a, w, sh, 1, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    reps = 20
    e0.record()
    for _ in range(reps):
        fused.linear(a, w, sh, 1, out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tot += us
    h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:10]
    print("%-28s %-18s %6d x %4d x %3d  %7.1f us  %6.1f TFLOP/s  %s" % (tag, name, rows, k, n, us, 2.0 * rows * k * n / us / 1e6, h), flush=True)
print("%-28s total %.1f us" % (tag, tot), flush=True)
