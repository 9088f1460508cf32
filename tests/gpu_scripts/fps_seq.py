"""16384-point D-FPS against the oracle + timing, on uniform / ray-cast / duplicated / lattice / all-equal / outlier clouds.
The library runs the multi-pick sampler (csrc/fps_seq.hip); DET6D_EXPERIMENTS_LIB=1 DET6D_FPS_SEQ=0 selects the one-pick
wave-skip sampler of rounds 2-3 (csrc/fps_cells.hip, experiments build only).  python tests/gpu_scripts/fps_seq.py [quick]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from de6d_amd import _lib as L
from de6d_amd.ops import fused
from oracle import ops as oops
from tests.util import make_batch, beam_batch

def timeouts():
    return 0

def run(name, xyz, m, check=2, reps=5):
    b, n, _ = xyz.shape
    x = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
    idx = torch.zeros((b, m), dtype=torch.int32, device='cuda')
    temp = fused.fps_workspace(b, n)
    fused.fps_fused(x, 0, n, m, None, 1.0, idx, 0, temp=temp); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fused.fps_fused(x, 0, n, m, None, 1.0, idx, 0, temp=temp)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    ok = None
    if check:
        ref = oops.fps(xyz[:check], m)
        got = idx.cpu().numpy()[:check]
        ok = np.array_equal(got, ref)
        if not ok:
            bad = np.argwhere(got != ref)
            print('   first mismatch at', bad[0], got[tuple(bad[0])], ref[tuple(bad[0])], 'of', len(bad))
    print("%-28s b=%d n=%d m=%d: %.3f ms (%.3f us/round) exact=%s timeouts=%d" % (name, b, n, m, best * 1e3, best * 1e6 / m, ok, timeouts()), flush=True)
    return ok

print('DET6D_FPS_SEQ =', os.environ.get('DET6D_FPS_SEQ'))
quick = 'quick' in sys.argv
n = 16384
good = True
u = np.ascontiguousarray(make_batch(1, 8, n, dup_frac=0.05)[..., :3])
good &= bool(run('uniform dup 5%', u[:1], 64, check=1, reps=1))
good &= bool(run('uniform dup 5%', u, 4096))
if not quick:
    bm = np.ascontiguousarray(beam_batch(3, 8, n)[..., :3])
    good &= bool(run('ray-cast', bm, 4096))
    d = u.copy(); d[:, n // 2:] = d[:, :n - n // 2]
    good &= bool(run('every point twice', d, 4096))
    lat = np.random.default_rng(0).integers(0, 12, (2, n, 3)).astype(np.float32)
    good &= bool(run('lattice (exact ties)', lat, 2048))
    good &= bool(run('all equal', np.ones((2, n, 3), np.float32), 64))
    out = u[:2].copy(); out[0, 5] = (1e4, -1e4, 50); out[1, 100] = (-3e3, 2e3, -70)
    good &= bool(run('outliers', out, 1024))
    good &= bool(run('b=32', np.ascontiguousarray(make_batch(9, 32, n)[..., :3]), 4096, check=2, reps=3))
    good &= bool(run('m=n', u[:1], n, check=1, reps=1))
# 4096-point scenes (the second layer's d-fps): the same kernel with 4 points per lane
n4 = 4096
u4 = np.ascontiguousarray(make_batch(21, 8, n4, dup_frac=0.05)[..., :3])
good &= bool(run('4096: uniform dup 5%', u4, 512))
d4 = u4.copy(); d4[:, n4 // 2:] = d4[:, :n4 - n4 // 2]
if not quick:
    good &= bool(run('4096: ray-cast', np.ascontiguousarray(beam_batch(5, 8, n4)[..., :3]), 512))
    good &= bool(run('4096: every point twice', d4, 1024))
    good &= bool(run('4096: lattice (exact ties)', np.random.default_rng(1).integers(0, 8, (2, n4, 3)).astype(np.float32), 700))
    good &= bool(run('4096: all equal', np.ones((2, n4, 3), np.float32), 64))
    good &= bool(run('4096: m=n', u4[:1], n4, check=1, reps=1))

if not quick:        # tiny pick counts: the first rounds on their own (short lists, one pick per round)
    for mm in (1, 2, 3, 33):
        good &= bool(run('m=%d' % mm, u[:3], mm, check=3, reps=1))
        good &= bool(run('4096: m=%d' % mm, u4[:3], mm, check=3, reps=1))
print('ALL EXACT' if good else 'MISMATCH')
sys.exit(0 if good else 1)
