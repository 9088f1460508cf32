"""where does the per-launch fixed cost of the GEMM go?  durations at tiny K, different row counts, N"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from de6d_amd.ops import fused
def t(r, k, n, reps=20):
    a = torch.randn((r, (k + 3) // 4 * 4), device='cuda'); w = torch.randn((k, n), device='cuda') / max(k, 1) ** 0.5
    sh = torch.randn((n,), device='cuda'); out = torch.empty((r, n), device='cuda')
    for _ in range(3): fused.linear(a, w, sh, 1, out, k=k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fused.linear(a, w, sh, 1, out, k=k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("rows %7d K %4d N %4d  %7.1f us   out %.0f MB  tiles %d" % (r, k, n, us, r * n * 4 / 1e6, (r // 128) * ((n + 127) // 128)))
for r in (32768, 65536, 131072, 262144):
    t(r, 16, 256)
t(131072, 16, 128); t(131072, 16, 512); t(131072, 16, 1024)
t(131072, 32, 256); t(131072, 64, 256); t(131072, 128, 256); t(131072, 256, 256); t(131072, 512, 256)
t(65536, 512, 1024); t(65536, 16, 1024)
print('--- pooled epilogue (ns = 32, masked), tiny output')
def tp(r, k, n, reps=20):
    a = torch.randn((r, (k + 3) // 4 * 4), device='cuda'); w = torch.randn((k, n), device='cuda') / max(k, 1) ** 0.5
    sh = torch.randn((n,), device='cuda'); out = torch.empty((r // 32, n), device='cuda')
    cnt = torch.ones((r // 32,), dtype=torch.int32, device='cuda')
    for _ in range(3): fused.linear(a, w, sh, 1, out, k=k, cnt=cnt, pool=32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fused.linear(a, w, sh, 1, out, k=k, cnt=cnt, pool=32)
    e1.record(); torch.cuda.synchronize()
    print("rows %7d K %4d N %4d  %7.1f us (pooled)" % (r, k, n, e0.elapsed_time(e1) / reps * 1e3))
tp(65536, 16, 1024); tp(65536, 32, 1024); tp(65536, 512, 1024); tp(65536, 16, 256); tp(131072, 16, 256); tp(131072, 256, 256)
