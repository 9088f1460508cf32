"""does the largest layer depend on where its input lives?  same launch with 1 vs 4 rotating input buffers,
random vs half-zero (post-ReLU-like) data"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from de6d_amd.ops import fused
r, k, n = 65536, 512, 1024
w = torch.randn((k, n), device='cuda') / k ** 0.5; sh = torch.randn((n,), device='cuda')
out = torch.empty((r // 32, n), device='cuda'); cnt = torch.ones((r // 32,), dtype=torch.int32, device='cuda')
def run(bufs, tag, reps=12):
    for a in bufs: fused.linear(a, w, sh, 1, out, cnt=cnt, pool=32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(reps): fused.linear(bufs[i % len(bufs)], w, sh, 1, out, cnt=cnt, pool=32)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("%-42s %7.1f us  %.1f TF" % (tag, us, 2.0 * r * k * n / us / 1e6))
rnd = [torch.randn((r, k), device='cuda') for _ in range(4)]
relu = [torch.relu(x) for x in rnd]
zeros = [torch.zeros((r, k), device='cuda') for _ in range(4)]
run(rnd[:1], 'random, one buffer')
run(rnd, 'random, four rotating buffers (536 MB)')
run(relu[:1], 'relu(random), one buffer')
run(relu, 'relu(random), four rotating buffers')
run(zeros[:1], 'zeros, one buffer')
