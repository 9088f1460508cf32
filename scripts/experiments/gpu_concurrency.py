"""how many kernels from different streams run at once? (NOT a result)  torch.cuda._sleep = one-workgroup spin kernel"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', sys.argv[1] if len(sys.argv) > 1 else '24')
import torch
x = torch.zeros(1, device='cuda')
cycles = 200000   # ~100 us
torch.cuda._sleep(cycles); torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(cycles); torch.cuda.synchronize(); one = time.perf_counter() - t0
print("one sleep kernel: %.1f us" % (one * 1e6))
for ns in (1, 2, 4, 8, 12, 16, 20, 22, 24, 32):
    st = [torch.cuda.Stream() for _ in range(ns)]
    reps = 20
    for s in st:
        with torch.cuda.stream(s): torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for s in st:
            with torch.cuda.stream(s): torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%2d streams x %d kernels: %.2f ms  -> effective concurrency %.1f" % (ns, reps, dt * 1e3, ns * reps * one / dt))
print("--- captured graphs of 20 sleep kernels per stream ---")
for ns in (1, 4, 8, 12, 16, 18, 20, 21, 22):
    st = [torch.cuda.Stream() for _ in range(ns)]
    graphs = []
    for s in st:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): torch.cuda._sleep(cycles)
        graphs.append(g)
    def run():
        for s, g in zip(st, graphs):
            with torch.cuda.stream(s): g.replay()
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%2d streams x 40 kernels (graphs): %.2f ms  -> effective concurrency %.1f (per-kernel slot %.1f us)" % (ns, dt * 1e3, ns * 40 * one / dt, dt / 40 * 1e6))
