"""round-4 debugging aid: what in bench.py's context makes latency_b1 read 5.5 ms when the stand-alone script reads 3.3"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import synth_points
from de6d_amd.runtime import load_config, build_model, GraphedDet6D, ScenePipeline

cfg = load_config('kitti_models/det6d_car.yaml')
model = build_model(cfg, seed=1234, device='cuda')
n = 16384


def b1(tag):
    one = torch.from_numpy(synth_points(4242, 1, n, tilt=False, scene='uniform')).cuda()
    lat1 = GraphedDet6D(model, 1, n, points=one)
    lat1.launch(); lat1.finalize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        lat1.launch(); lat1.finalize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, 'latency_b1 ms: first %.3f min %.3f mean %.3f max %.3f' % (ts[0], min(ts), sum(ts) / len(ts), max(ts)), flush=True)


b1('fresh process')
pts8 = torch.from_numpy(synth_points(1, 8, n, tilt=False, scene='uniform')).cuda()
lat = GraphedDet6D(model, 8, n, points=pts8)
for _ in range(6):
    lat.launch(); lat.finalize()
b1('with a live 8-scene runner')
del lat
b1('after the 8-scene runner was freed')
pts32 = torch.from_numpy(synth_points(1, 32, n, tilt=False, scene='uniform')).cuda()
pipe = ScenePipeline(model, 32, n, n_main=16, group=1, prefetch=4, sampler_streams=6, points=pts32)
pipe.run(64)
torch.cuda.synchronize()
b1('with a live pipeline (idle)')
del pipe
torch.cuda.empty_cache()
b1('after the pipeline was freed')
