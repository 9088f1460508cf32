"""what-if (NOT a result): bench.py's pipeline with every sampler replaced by a copy of the indices it would produce
-> what farthest point sampling costs in the in-flight regime.  usage: gpu_whatif2.py [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from de6d_amd.ops import fused
from de6d_amd.runtime import load_config, build_model
_real, _rec = fused.fps_fused, {}
def rec_fps(xyz, lo, hi, m, scores, gamma, idx_out, idx_offset, temp=None):
    _real(xyz, lo, hi, m, scores, gamma, idx_out, idx_offset, temp=temp)
    _rec[(lo, hi, m, scores is None)] = idx_out[:, idx_offset:idx_offset + m].clone()
fused.fps_fused = rec_fps
cfg = load_config('kitti_models/det6d_car.yaml')
model = build_model(cfg, seed=1234, device='cuda')
with torch.no_grad():
    model({'batch_size': 8, 'points': torch.from_numpy(bench.synth_points(1000, 8, 16384)).cuda()})
torch.cuda.synchronize()
def fake_fps(xyz, lo, hi, m, scores, gamma, idx_out, idx_offset, temp=None):
    ctl = fused.SAMPLER_SEGMENTS
    if ctl is not None and ctl.recording:
        ctl.add_sampler(xyz, lo, hi, m, scores, gamma, idx_out, idx_offset); return
    r = _rec[(lo, hi, m, scores is None)]
    idx_out[:, idx_offset:idx_offset + m] = r.repeat(idx_out.shape[0] // r.shape[0], 1)
mode = os.environ.get('WHATIF', 'nofps')
if 'nofps' in mode:
    fused.fps_fused = fake_fps
else:
    fused.fps_fused = _real
if 'nolinear' in mode:      # GEMM family stubbed out (outputs keep whatever the buffers hold): what everything else costs
    _orig = fused.L.call
    def call(name, *args):
        if name in ('det6d_linear', 'det6d_mlp_chain3', 'det6d_mlp_chain3_compact'):
            return 0
        return _orig(name, *args)
    fused.L.call = call
if 'halfk' in mode:          # linear launches with half the reduction length: how GEMM-bound is the pipeline?
    _lin = fused.linear
    def half_linear(a, w, shift, act, out, k=None, **kw):
        kk = w.shape[0] if k is None else k
        return _lin(a, w, shift, act, out, k=max(4, (kk // 2) // 4 * 4), **kw)
    fused.linear = half_linear
    import de6d_amd.pcdet.ops.pointnet2.pointnet2_batch.pointnet2_modules as _pm
if 'nocompact' in mode:
    _orig2 = fused.L.call
    def call2(name, *args):
        if name == 'det6d_compact_groups':
            return 0
        return _orig2(name, *args)
    fused.L.call = call2
bench.main()
