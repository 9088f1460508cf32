"""how many hardware queues do high-priority streams get? (NOT a result)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from de6d_amd.ops import fused as F
from bench import synth_points
B, N, M = 8, 16384, 4096
pts = torch.from_numpy(synth_points(1000, B, N)).cuda()
rows, xyz = F.pack_points(pts, 4)
xyz = xyz.view(B, N, 3)
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
for prio in (0, -1, 1):
    for ns in (1, 2, 4, 8, 12):
        try:
            st = [torch.cuda.Stream(priority=prio) for _ in range(ns)]
        except Exception as e:
            print("priority", prio, "failed:", e); break
        idxs = [torch.empty((B, M), dtype=torch.int32, device='cuda') for _ in st]
        temps = [torch.empty((B, N), device='cuda') for _ in st]
        for i, s in enumerate(st):
            with torch.cuda.stream(s): F.fps_fused(xyz, 0, N, M, None, 1.0, idxs[i], 0, temp=temps[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(2):
            for i, s in enumerate(st):
                with torch.cuda.stream(s): F.fps_fused(xyz, 0, N, M, None, 1.0, idxs[i], 0, temp=temps[i])
        torch.cuda.synchronize()
        print("priority %2d, %2d streams x 2 samplers: %.2f ms (ids %s)" % (prio, ns, (time.perf_counter() - t0) * 1e3, sorted(set(s.cuda_stream for s in st))[:3]))
