"""what-if (NOT a result): the hoisted first-layer sampler replaced by (a copy of its picks) + a workgroup per scene that only HOLDS
a register / LDS footprint for the sampler's duration (experiments build: det6d_dbg_occupy) -> what the sampler costs the
pipeline by FOOTPRINT alone, and what a smaller footprint would buy.
usage: DET6D_EXPERIMENTS_LIB=1 whatif_occupancy.py <regs: 0 = no occupier | 16 | 40 | 56 | 80 | 96> <lds bytes> <usec> [bench args]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
os.environ['DET6D_EXPERIMENTS_LIB'] = '1'
import torch
regs, lds, usec = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sys.argv = [sys.argv[0]] + sys.argv[4:]
import bench
from de6d_amd.ops import fused
from de6d_amd import _lib as L
_real, _cache = fused.fps_fused, {}
sink = None


def cached(xyz, lo, hi, m, scores, gamma, idx_out, idx_offset, temp=None, idx_bias=0):
    global sink
    take = scores is None and temp is not None and not torch.cuda.is_current_stream_capturing() and hi - lo == 16384
    if not take:
        return _real(xyz, lo, hi, m, scores, gamma, idx_out, idx_offset, temp=temp, idx_bias=idx_bias)
    key = (xyz.data_ptr(), lo, hi, m, idx_out.data_ptr(), idx_offset, idx_out.shape[0])
    if key not in _cache:
        _real(xyz, lo, hi, m, scores, gamma, idx_out, idx_offset, temp=temp, idx_bias=idx_bias)
        _cache[key] = idx_out[:, idx_offset:idx_offset + m].clone()
        return
    if regs > 0:
        if sink is None:
            sink = torch.zeros((4,), device='cuda')
            L.lib().det6d_dbg_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.lib().det6d_dbg_occupy(xyz.shape[0], regs, lds, usec, L.ptr(sink), L.stream_ptr())
    idx_out[:, idx_offset:idx_offset + m].copy_(_cache[key], non_blocking=True)


fused.fps_fused = cached
bench.main()
