"""CU-mask experiment (NOT a result): samplers on streams masked to R reserved CUs, GEMMs on streams masked to the rest;
also: does a captured graph launched on a masked stream honour the mask?"""
import os, sys, time, ctypes
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from de6d_amd.ops import fused as F
from bench import synth_points
hip = ctypes.CDLL('libamdhip64.so')
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << (i - 32 * w) for i in bits if 32 * w <= i < 32 * w + 32) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
PER_XCD = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reserved = sorted(32 * x + (x + 8 * k) % 32 for x in range(8) for k in range(PER_XCD))
rest = [i for i in range(256) if i not in reserved]
B, N, M = 8, 16384, 4096
pts = torch.from_numpy(synth_points(1000, B, N)).cuda()
rows, xyz = F.pack_points(pts, 4)
xyz = xyz.view(B, N, 3)
R, K, C = 65536, 512, 1024
a = torch.relu(torch.randn(R, K, device='cuda')); w = torch.randn(K, C, device='cuda') * 0.05; sh = torch.zeros(C, device='cuda')
outs = [torch.empty(R, C, device='cuda') for _ in range(4)]
gs = [masked_stream(rest) for _ in range(4)]
fs = [masked_stream(reserved) for _ in range(6)]
plain = [torch.cuda.Stream() for _ in range(4)]
idxs = [torch.empty((B, M), dtype=torch.int32, device='cuda') for _ in fs]
def gemm_loop(streams, reps):
    for i in range(reps):
        for si, s in enumerate(streams):
            with torch.cuda.stream(s):
                F.linear(a, w, sh, 1, outs[si])
def fps_loop(streams, reps, evs):
    for i in range(reps):
        for si, s in enumerate(streams):
            with torch.cuda.stream(s):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s); F.fps_fused(xyz, 0, N, M, None, 1.0, idxs[si], 0); e1.record(s); evs.append((e0, e1))
gemm_loop(gs, 2); gemm_loop(plain, 2); ev = []; fps_loop(fs[:1], 1, ev); torch.cuda.synchronize()
for name, st in (("unmasked", plain), ("masked to %d CUs" % len(rest), gs)):
    t0 = time.perf_counter(); gemm_loop(st, 40); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("GEMM alone, %s: %.1f TF" % (name, 160 * 2.0 * R * K * C / dt / 1e12))
ev = []; fps_loop(fs[:1], 3, ev); torch.cuda.synchronize()
print("FPS alone on %d reserved CUs: %.3f ms" % (len(reserved), np.mean([a_.elapsed_time(b_) for a_, b_ in ev])))
def trial(label, gstreams, fstreams):
    for nf in (1, 4):
        ev = []
        t0 = time.perf_counter()
        gemm_loop(gstreams, 40); fps_loop(fstreams[:nf], 4, ev)
        for s in gstreams: s.synchronize()
        dtg = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("%-58s %d sampler stream(s): GEMM %.1f TF, sampler %.3f ms avg (max %.3f)" % (
            label, nf, 160 * 2.0 * R * K * C / dtg / 1e12, np.mean([a_.elapsed_time(b_) for a_, b_ in ev]), max(a_.elapsed_time(b_) for a_, b_ in ev)))
hi = [torch.cuda.Stream(priority=-1) for _ in range(6)]
lo = [torch.cuda.Stream() for _ in range(6)]
trial("GEMM masked / sampler masked to reserved", gs, fs)
trial("GEMM masked / sampler unmasked high priority", gs, hi)
trial("GEMM masked / sampler unmasked normal priority", gs, lo)
trial("GEMM unmasked / sampler unmasked high priority", plain, hi)
trial("GEMM unmasked / sampler unmasked normal priority", plain, lo)
# graph on a masked stream
small = masked_stream(reserved)
g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream()
with torch.cuda.stream(cap):
    F.linear(a, w, sh, 1, outs[0])
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=cap):
        for _ in range(4): F.linear(a, w, sh, 1, outs[0])
for name, st in (("plain stream", plain[0]), ("stream masked to %d CUs" % len(reserved), small)):
    with torch.cuda.stream(st):
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        print("graph of 4 GEMMs replayed on %s: %.2f ms per replay" % (name, (time.perf_counter() - t0) / 5 * 1e3))
