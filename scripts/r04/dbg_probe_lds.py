import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from de6d_amd import _lib as L
out = torch.zeros(64 * 1024, dtype=torch.int32, device='cuda')
L.lib().det6d_dbg_probe_lds.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
print('rc', L.lib().det6d_dbg_probe_lds(L.ptr(out), 64, L.stream_ptr()))
torch.cuda.synchronize()
v, c = torch.unique(out, return_counts=True)
print('uninitialised LDS words seen:', [(hex(int(a) & 0xffffffff), int(b)) for a, b in zip(v[:8], c[:8])], 'distinct', len(v))
