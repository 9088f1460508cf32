"""interference experiment (NOT a result): SA1 farthest point sampling (8 scenes) on one stream against a stream
of large fp32 GEMMs on others: sampler duration alone / under load, GEMM rate alone / beside k samplers"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from de6d_amd.ops import fused as F
from bench import synth_points
B, N, M = 8, 16384, 4096
pts = torch.from_numpy(synth_points(1000, B, N)).cuda()
rows, xyz = F.pack_points(pts, 4)
xyz = xyz.view(B, N, 3)
R, K, C = 65536, 512, 1024
a = torch.relu(torch.randn(R, K, device='cuda')); w = torch.randn(K, C, device='cuda') * 0.05; sh = torch.zeros(C, device='cuda')
outs = [torch.empty(R, C, device='cuda') for _ in range(4)]
def gemm_loop(streams, reps):
    for i in range(reps):
        for si, s in enumerate(streams):
            with torch.cuda.stream(s):
                F.linear(a, w, sh, 1, outs[si])
def fps_loop(streams, reps, evs=None):
    for i in range(reps):
        for si, s in enumerate(streams):
            with torch.cuda.stream(s):
                idx = idxs[si]
                if evs is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                F.fps_fused(xyz, 0, N, M, None, 1.0, idx, 0)
                if evs is not None:
                    e1.record(s); evs.append((e0, e1))
gs = [torch.cuda.Stream() for _ in range(4)]
fs = [torch.cuda.Stream() for _ in range(6)]
idxs = [torch.empty((B, M), dtype=torch.int32, device='cuda') for _ in fs]
gemm_loop(gs, 2); fps_loop(fs[:1], 1); torch.cuda.synchronize()
# GEMM alone
t0 = time.perf_counter(); gemm_loop(gs, 40); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("GEMM alone: %.1f TF  (%.1f us per GEMM)" % (160 * 2.0 * R * K * C / dt / 1e12, dt / 160 * 1e6))
ev = []; fps_loop(fs[:1], 3, ev); torch.cuda.synchronize()
print("FPS alone: %.3f ms" % np.mean([a_.elapsed_time(b_) for a_, b_ in ev]))
import ctypes
def kernel_clock():
    buf = (ctypes.c_ulonglong * 128)()
    F.L.lib().det6d_dbg_fps_clock(buf)
    return [(buf[2 * i + 1] - buf[2 * i]) / 100e3 for i in range(B)]   # ms inside the kernel, per scene
print("in-kernel ms per scene (alone):", ["%.2f" % v for v in kernel_clock()])
for nf in (1, 2, 3, 6):
    ev = []
    t0 = time.perf_counter()
    gemm_loop(gs, 40)
    fps_loop(fs[:nf], 4, ev)
    for s in gs: s.synchronize()
    dtg = time.perf_counter() - t0
    torch.cuda.synchronize()
    dta = time.perf_counter() - t0
    print("%d sampler stream(s) beside the GEMMs: GEMM %.1f TF (done after %.1f ms), sampler %.3f ms avg, all done after %.1f ms" % (
        nf, 160 * 2.0 * R * K * C / dtg / 1e12, dtg * 1e3, np.mean([a_.elapsed_time(b_) for a_, b_ in ev]), dta * 1e3))
    print("   in-kernel ms per scene (last launch):", ["%.2f" % v for v in kernel_clock()])
