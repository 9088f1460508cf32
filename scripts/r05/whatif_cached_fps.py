"""what-if (NOT a result): bench.py's worker with the hoisted D-FPS launches of every group replaced, after their first run, by a
copy of the picks they produced (the groups' inputs are resident and never change, so every later result is still correct and
the self-check passes) -> what the input-only samplers cost the pipeline today.
usage: whatif_cached_fps.py sa1|chain|none [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
import torch
mode = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
import bench
from de6d_amd.ops import fused
_real, _cache = fused.fps_fused, {}


def cached(xyz, lo, hi, m, scores, gamma, idx_out, idx_offset, temp=None, idx_bias=0):
    take = scores is None and temp is not None and not torch.cuda.is_current_stream_capturing() and (mode == 'chain' or (mode == 'sa1' and hi - lo == 16384))
    if not take:
        return _real(xyz, lo, hi, m, scores, gamma, idx_out, idx_offset, temp=temp, idx_bias=idx_bias)
    key = (xyz.data_ptr(), lo, hi, m, idx_out.data_ptr(), idx_offset, idx_out.shape[0])
    if key not in _cache:
        _real(xyz, lo, hi, m, scores, gamma, idx_out, idx_offset, temp=temp, idx_bias=idx_bias)
        _cache[key] = idx_out[:, idx_offset:idx_offset + m].clone()
        return
    idx_out[:, idx_offset:idx_offset + m].copy_(_cache[key], non_blocking=True)


if mode != 'none':
    fused.fps_fused = cached
    import de6d_amd.runtime as rt
bench.main()
