"""Parity on EXACTLY what bench.py times ("parity on what is timed"): kitti_models/det6d_car.yaml at
batch 8 x 16384 points through the two-stage pipeline (Det6DGroup of 4 passes: grouped 32-scene first sampler + captured
graph segments; ScenePipeline around it), on compact rows and on the reference's dense rows, against oracle/model.py bit
for bit; the sloped and the 65536-point configurations through captured graphs; and the scene-sharded HIP engine on two
rank processes."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.test_model_gpu import check, flat_points
from tests.util import make_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_of(cfg, model, pts_np, b):
    from oracle import model as omodel
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    return omodel.forward(cfg.MODEL, sd, pts_np, b)


def test_pipelined_tests_run_on_the_benchmarked_queues():
    """round-4 review: the ScenePipeline-vs-oracle tests below ran on the default 4 hardware queues (22 streams aliased and
    serialised) while bench.py times 24.  tests/conftest.py exports GPU_MAX_HW_QUEUES=24 before anything initialises HIP;
    and a process that initialised HIP FIRST with the default queues gets a RuntimeError from ScenePipeline, not a warning."""
    import de6d_amd
    assert de6d_amd.HW_QUEUES_AT_IMPORT is None and int(os.environ['GPU_MAX_HW_QUEUES']) >= 24
    code = (
        "import os, sys; os.environ.pop('GPU_MAX_HW_QUEUES', None); sys.path.insert(0, %r)\n"
        "import torch; torch.zeros(1, device='cuda'); torch.cuda.synchronize()      # the caller's own context first\n"
        "import de6d_amd; assert de6d_amd.HW_QUEUES_AT_IMPORT == 4\n"
        "from de6d_amd.runtime import load_config, build_model, ScenePipeline\n"
        "cfg = load_config('synthetic_models/det6d_tiny.yaml'); model = build_model(cfg, seed=3, device='cuda')\n"
        "try:\n"
        "    ScenePipeline(model, 2, 2048, n_main=16, group=4, prefetch=4, sampler_streams=6)\n"
        "except RuntimeError as e:\n"
        "    assert 'GPU_MAX_HW_QUEUES' in str(e); print('raised')\n"
        "pipe = ScenePipeline(model, 2, 2048, n_main=2, group=1, prefetch=1, sampler_streams=1)   # 3 streams fit 4 queues\n"
        "print('small ok')\n" % ROOT)
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'raised' in out.stdout and 'small ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_bench_group_full_size_vs_oracle(oracle_ops):
    """the benchmarked shape: Det6DGroup(k=4), batch 8 x 16384, four DIFFERENT batches; every pass against the oracle:
    sampled points of every level, features, logits, boxes, kept detections and their order — bit-exact"""
    from de6d_amd.runtime import load_config, build_model, Det6DGroup
    cfg = load_config('kitti_models/det6d_car.yaml')
    model = build_model(cfg, seed=1234, device='cuda')
    b, n, k = 8, 16384, 4
    batches_np = [flat_points(make_batch(7000 + 100 * j, b, n)) for j in range(k)]
    batches = [torch.from_numpy(p).cuda() for p in batches_np]
    group = Det6DGroup(model, b, n, k, torch.cuda.Stream(), points=batches,
                       main_streams=[torch.cuda.Stream() for _ in range(k)])
    for rep in range(2):                       # the second replay must give the same answer as the first
        passes = group.launch()
        preds = [r.finalize() for r in passes]
    torch.cuda.synchronize()
    for j in (0, k - 1, 1, 2):                 # first and last pass first
        ref = oracle_of(cfg, model, batches_np[j], b)
        check(passes[j].batch_dict, preds[j], ref, b)
        assert sum(len(p['pred_scores']) for p in preds[j]) > 0


def test_bench_group_full_size_dense_rows():
    """the same test on the reference's dense (centre x nsample) rows: bench.py's `dense_rows` leg
    (DET6D_DENSE_ROWS is read at import: child process)"""
    if os.environ.get('DET6D_DENSE_ROWS'):
        pytest.skip('already the dense-rows child')
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-m', 'gpu', '-k',
                          'test_bench_group_full_size_vs_oracle or test_scene_pipeline'],
                         env=dict(os.environ, DET6D_DENSE_ROWS='1'), cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert 'passed' in out.stdout


@pytest.mark.parametrize("cfg_name,b,n,tilt", [('slopedkitti_models/det6d_car.yaml', 8, 16384, True),
                                               ('synthetic_models/det6d_65536.yaml', 2, 65536, False),
                                               ('kitti_models/det6d_3class.yaml', 4, 16384, False)])
def test_other_baseline_configs_through_captured_graphs(oracle_ops, cfg_name, b, n, tilt):
    """BASELINE configs 3, 5 and 4 (per-GPU share: 32 scenes / 8 GPUs) through GraphedDet6D, replayed twice with two
    different batches, against the oracle"""
    from de6d_amd.runtime import load_config, build_model, GraphedDet6D
    cfg = load_config(cfg_name)
    model = build_model(cfg, seed=77, device='cuda')
    runner = GraphedDet6D(model, b, n)
    for seed in (8100, 8200):
        pts_np = flat_points(make_batch(seed, b, n, tilt=tilt))
        preds = runner.launch(torch.from_numpy(pts_np).cuda()).finalize()
        torch.cuda.synchronize()
        ref = oracle_of(cfg, model, pts_np, b)
        check(runner.batch_dict, preds, ref, b)
    if tilt:
        assert (ref['batch_box_preds'][:, 7] != 0).any()      # the pitch branch of the ground-aware decoder occurred


def test_scene_pipeline_stream_equals_eager(oracle_ops):
    """ScenePipeline as bench.py drives it (16 main + 6 sampler streams, groups of 4, 4 groups ahead) on the full-size
    model: a stream of steps whose length is not a multiple of the group size, distinct resident batches, every finalised
    step compared with the eager model; then the same pipeline fed from the host (the --h2d route)"""
    from de6d_amd.runtime import load_config, build_model, ScenePipeline
    cfg = load_config('kitti_models/det6d_car.yaml')
    model = build_model(cfg, seed=1234, device='cuda')
    b, n = 8, 16384
    batches = [torch.from_numpy(flat_points(make_batch(9000 + 50 * j, b, n))).cuda() for j in range(3)]
    with torch.no_grad():
        eager = [model({'batch_size': b, 'points': p})[0] for p in batches]
    torch.cuda.synchronize()
    pipe = ScenePipeline(model, b, n, n_main=16, group=4, prefetch=4, sampler_streams=6, points=batches)
    which = {id(r): i % len(batches) for i, r in enumerate(pipe.passes)}
    seen = []

    def on_done(step, r, preds):
        want = eager[which[id(r)]]
        for g, e in zip(preds, want):
            assert torch.equal(g['pred_boxes'], e['pred_boxes']) and torch.equal(g['pred_scores'], e['pred_scores'])
            assert torch.equal(g['pred_labels'], e['pred_labels'])
        seen.append(step)
    assert pipe.run(70, on_done=on_done) == 70
    assert seen == list(range(70))
    assert pipe.run(3, on_done=on_done) == 3          # shorter than one group

    # host-fed: a callable writes each pass's input on the sampler stream
    feed_order = []

    def feed(r):
        j = len(feed_order) % len(batches)
        feed_order.append(j)
        r.points.copy_(batches[j], non_blocking=True)
    pipe2 = ScenePipeline(model, b, n, n_main=8, group=2, prefetch=2, sampler_streams=3, points=None,
                          main_streams=pipe.main_streams[:8], samplers=pipe.sampler_streams[:3])
    step_feed = []

    def on_done2(step, r, preds):
        j = step_feed[step]
        for g, e in zip(preds, eager[j]):
            assert torch.equal(g['pred_boxes'], e['pred_boxes']) and torch.equal(g['pred_scores'], e['pred_scores'])
    feed_order.clear()
    # passes are fed in launch order == step order
    n_steps = 21
    step_feed = [i % len(batches) for i in range(n_steps)]
    assert pipe2.run(n_steps, feed=feed, on_done=on_done2) == n_steps

    # coalesced passes (bench.py's default: consecutive batches share a pass): every STEP equals the one-batch eager result
    for merge, shape in ((2, dict(n_main=8, group=2, prefetch=2)), (4, dict(n_main=6, group=1, prefetch=3))):
        inputs = ScenePipeline.coalesce(batches, merge)
        pipe3 = ScenePipeline(model, b, n, sampler_streams=3, points=inputs, merge=merge,
                              main_streams=pipe.main_streams[:shape['n_main']], samplers=pipe.sampler_streams[:3], **shape)
        assert all(r.batch_size == b * merge for r in pipe3.passes)
        which3 = {id(r): i % len(inputs) for i, r in enumerate(pipe3.passes)}
        slot = {}
        seen3 = []

        def on_done3(step, r, preds):
            j = slot.get(id(r), 0)
            slot[id(r)] = (j + 1) % merge
            want = eager[(which3[id(r)] * merge + j) % len(batches)]
            assert len(preds) == b
            for g, e in zip(preds, want):
                assert torch.equal(g['pred_boxes'], e['pred_boxes']) and torch.equal(g['pred_scores'], e['pred_scores'])
                assert torch.equal(g['pred_labels'], e['pred_labels'])
            seen3.append(step)
        n3 = 10 * merge * pipe3.k + 1      # not a multiple of the pass: the last pass reports its first batch only
        assert pipe3.run(n3, on_done=on_done3) == n3
        assert seen3 == list(range(n3))
        del pipe3


def test_device_iou_equals_host_entry():
    """det6d_boxes_iou_bev (device) and det6d_boxes_iou_bev_cpu (the host entry of the reference interface) share
    include/det6d_geom.h: identical bits"""
    from de6d_amd.ops import iou3d_nms_hip
    from tests.util import random_boxes
    a = torch.from_numpy(random_boxes(21, 300))
    bx = torch.from_numpy(random_boxes(22, 130))
    host = torch.zeros((300, 130))
    iou3d_nms_hip.boxes_iou_bev_cpu(a, bx, host)
    dev = torch.zeros((300, 130), device='cuda')
    iou3d_nms_hip.boxes_iou_bev_gpu(a.cuda(), bx.cuda(), dev)
    assert torch.equal(dev.cpu(), host)


# ---- two rank processes: the scene-sharded HIP engine ---------------------------------------------------------------

_RANK_SCRIPT = r'''
import os, sys, json
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ['DET6D_ROOT'])
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
ndev = torch.cuda.device_count()
torch.cuda.set_device(rank % ndev)
# one device per rank -> RCCL; ranks sharing a device (1-GPU box) -> gloo carries the (host-side) merge
dist.init_process_group('nccl' if ndev >= world else 'gloo', rank=rank, world_size=world)
from de6d_amd import parallel
from de6d_amd.runtime import load_config, build_model, GraphedDet6D
from tests.test_model_gpu import flat_points
from tests.util import make_batch
cfg = load_config('synthetic_models/det6d_tiny.yaml')
model = build_model(cfg, seed=3, device='cuda')
num_scenes, n, b = 11, 2048, 2
mine = parallel.scene_shard(num_scenes, rank, world)          # DistributedSampler(shuffle=False) order
runner = GraphedDet6D(model, b, n)
results = []
for i in range(0, len(mine), b):
    ids = mine[i:i + b]
    while len(ids) < b:
        ids = ids + [ids[-1]]
    pts = flat_points(np.stack([make_batch(500 + s, 1, n)[0] for s in ids], 0))
    preds = runner.launch(torch.from_numpy(pts).cuda()).finalize()
    for s, p in list(zip(ids, preds))[:len(mine[i:i + b])]:
        results.append({'scene': s, 'boxes': p['pred_boxes'].cpu().numpy(), 'scores': p['pred_scores'].cpu().numpy()})
merged = parallel.gather_detections(results, num_scenes)
# the optional training-side collective (north star: "RCCL all-reduce over xGMI wired only for the optional training grad
# step"): the model's whole gradient as ONE flat bucket, one all-reduce.  Under RCCL when every rank has a device of its
# own; on a one-GPU box the bucket is reduced over gloo from host copies of the same gradients.
params = [p for p in model.parameters()]
for i, p in enumerate(params):
    p.grad = torch.full_like(p, float(rank + 1) * (1.0 + (i % 7)))
if ndev >= world:
    nred = parallel.allreduce_gradients(params)
    got = [p.grad for p in params]
else:
    host = [torch.nn.Parameter(p.detach().cpu()) for p in params]
    for h, p in zip(host, params):
        h.grad = p.grad.cpu()
    nred = parallel.allreduce_gradients(host)
    got = [h.grad for h in host]
assert nred == sum(p.numel() for p in params)
mean_rank = (world + 1) / 2.0
for i, g in enumerate(got):
    assert torch.all(g == mean_rank * (1.0 + (i % 7))), i
if rank == 0:
    np.savez(os.environ['DET6D_OUT'], order=np.array([m['scene'] for m in merged]),
             **{'boxes_%d' % m['scene']: m['boxes'] for m in merged}, **{'scores_%d' % m['scene']: m['scores'] for m in merged})
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_hip_engine_scene_sharding(tmp_path, oracle_ops):
    """core/tools/test.py:137-143 + DistributedSampler(shuffle=False) + merge_results_dist
    (core/pcdet/datasets/__init__.py:27-70, common_utils.py:212-233): two rank processes run the HIP engine on their
    scene shard (scene_shard) and merge with gather_detections; the merged list is in scene order and every scene's
    detections equal the oracle's.  With two or more devices the ranks use one device each over RCCL; on a one-GPU box
    both ranks share cuda:0 and the host-side merge goes over gloo (the data path has no collective either way)."""
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    out_file = str(tmp_path / 'merged.npz')
    script = tmp_path / 'rank.py'
    script.write_text(_RANK_SCRIPT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   DET6D_ROOT=ROOT, DET6D_OUT=out_file, HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    z = np.load(out_file)
    np.testing.assert_array_equal(z['order'], np.arange(11))
    from de6d_amd.runtime import load_config, build_model
    from oracle import model as omodel
    cfg = load_config('synthetic_models/det6d_tiny.yaml')
    model = build_model(cfg, seed=3)
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    for s in range(11):
        pts = flat_points(make_batch(500 + s, 1, 2048))
        ref = omodel.forward(cfg.MODEL, sd, pts, 1)['pred_dicts'][0]
        np.testing.assert_array_equal(z['boxes_%d' % s], ref['pred_boxes'])
        np.testing.assert_array_equal(z['scores_%d' % s], ref['pred_scores'])


def test_ray_cast_lidar_scenes_vs_oracle(oracle_ops):
    """bench.py's `--scene beam` legs: 64-ring ray-cast scenes (dense near the sensor: balls 0.3-0.9 full, many exact
    nsample-capped neighbourhoods) through the compact-row engine against the oracle's dense rows"""
    from de6d_amd.runtime import load_config, build_model
    from tests.util import beam_batch
    cfg = load_config('kitti_models/det6d_car.yaml')
    model = build_model(cfg, seed=1234, device='cuda')
    b, n = 2, 16384
    pts_np = flat_points(beam_batch(4100, b, n))
    bd = {'batch_size': b, 'points': torch.from_numpy(pts_np).cuda()}
    with torch.no_grad():
        pred, _ = model(bd)
    check(bd, pred, oracle_of(cfg, model, pts_np, b), b)


def test_bench_entry_point_checks_itself():
    """the worker of bench.py on a short run: one JSON line with the contract's keys, the steady-state timing description,
    `selfcheck: ok` (every pass compared with the eager model inside the bench), cold / latency keys"""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '8', '--warmup', '2', '--worker',
                          '--no-legs', '--no-roofline', '--cpu-scenes', '0', '--windows', '3', '--preroll', '1'],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'selfcheck', 'cold', 'latency', 'ranks_seen'):
        assert key in d, key
    assert d['selfcheck'] == 'ok' and d['steps'] == 8 and d['n_gpus'] == 1 and d['dtype'] == 'f32' and d['value'] > 0
    assert 'steady-state' in d['config']['timing'] and d['config']['points_per_scene'] == 16384
    assert d['config']['batches_per_pass'] == 8 and d['config']['scenes_per_pass'] == 64      # 8 steps = one coalesced pass


@pytest.mark.parametrize("scene,merge", [("beam", 4), ("uniform", 4), ("uniform", 10), ("beam", 10)])
def test_coalesced_32_scene_passes_through_scene_pipeline_vs_oracle(oracle_ops, scene, merge):
    """what bench.py's `value` (uniform scenes) and its `--scene beam` leg (ray-cast scenes, KITTI-like ball fill 0.3-0.9: the
    GEMM-bound regime) time: batch 8 x 16384, consecutive batches coalesced into 80-scene passes (merge 10: the bench's default
    from round 5 on) and 32-scene passes (merge 4: rounds 2-4, the `operating_points` legs) through ScenePipeline, every step
    against the ORACLE's dense rows bit for bit — the 64-row mlp_rows tiles, the 2048-workgroup chain grids and (ray-cast) the
    full-ball class-32 tiles of the compact lists only occur at these row counts"""
    from de6d_amd.runtime import load_config, build_model, ScenePipeline
    from tests.util import beam_batch
    cfg = load_config('kitti_models/det6d_car.yaml')
    model = build_model(cfg, seed=1234, device='cuda')
    b, n = 8, 16384
    make = beam_batch if scene == "beam" else make_batch
    batches_np = [flat_points(make(4300 + 20 * j, b, n)) for j in range(merge)]
    batches = [torch.from_numpy(p).cuda() for p in batches_np]
    inputs = ScenePipeline.coalesce(batches, merge)                      # one 32-scene pass input
    pipe = ScenePipeline(model, b, n, n_main=4, group=1, prefetch=2, sampler_streams=2, points=inputs, merge=merge)
    got = {}

    def on_done(step, r, preds):
        got[step] = [{k: v.clone() for k, v in p.items()} for p in preds]
    # the 32-scene cases run PACED (ScenePipeline.run headway, the bench's open-loop operating points): same results
    assert pipe.run(2 * merge, on_done=on_done, headway=2e-3 if merge == 4 else 0.0) == 2 * merge
    torch.cuda.synchronize()
    for j in range(merge):
        ref = oracle_of(cfg, model, batches_np[j], b)
        for step in (j, j + merge):                                     # every pass holds batches 0..3 in order
            for g, w in zip(got[step], ref['pred_dicts']):
                np.testing.assert_array_equal(g['pred_boxes'].cpu().numpy(), w['pred_boxes'])
                np.testing.assert_array_equal(g['pred_scores'].cpu().numpy(), w['pred_scores'])
                np.testing.assert_array_equal(g['pred_labels'].cpu().numpy(), w['pred_labels'])
        assert sum(len(p['pred_scores']) for p in got[j]) > 0
    # and the intermediate levels of one coalesced pass against the oracle run on all 32 scenes at once
    r = pipe.passes[0]
    r_pts = torch.cat(batches, 0).cpu().numpy()
    r_pts[:, 0] = np.repeat(np.arange(b * merge, dtype=np.float32), n)
    preds = r.finalize()
    check(r.batch_dict, preds, oracle_of(cfg, model, r_pts, b * merge), b * merge)


def test_cooperative_sampler_pipeline_under_load_reports_status(oracle_ops):
    """65536-point scenes (BASELINE config 5) through ScenePipeline while 16 other streams keep the chip full of GEMM
    workgroups: the cooperative sampler's parts must all get scheduled (ScenePipeline bounds the cooperative launches in
    flight), its sticky error word stays 0 (finalize() would raise FpsTimeout), results equal the eager model"""
    from de6d_amd.ops import fused
    from de6d_amd.runtime import load_config, build_model, ScenePipeline
    cfg = load_config('synthetic_models/det6d_65536.yaml')
    model = build_model(cfg, seed=77, device='cuda')
    b, n = 2, 65536
    assert fused.fps_is_cooperative(n) and not fused.fps_is_cooperative(16384)
    batches = [torch.from_numpy(flat_points(make_batch(8300 + 10 * j, b, n))).cuda() for j in range(2)]
    with torch.no_grad():
        eager = [model({'batch_size': b, 'points': p})[0] for p in batches]
    torch.cuda.synchronize()
    pipe = ScenePipeline(model, b, n, n_main=4, group=1, prefetch=2, sampler_streams=6, points=batches)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert len(pipe.sampler_streams) <= max(1, cus // (b * 4))       # bounded: 8 cooperative workgroups per launch here
    assert all(g._status_words for g in pipe.groups)          # every group watches its sampler's error word
    # background load: 16 streams of large GEMMs for the whole run
    x = torch.randn((8192, 1024), device='cuda')
    w = torch.randn((1024, 1024), device='cuda')
    load_streams = [torch.cuda.Stream() for _ in range(16)]
    stop = torch.zeros((), dtype=torch.int32)

    def load(k):
        for s in load_streams:
            with torch.cuda.stream(s):
                for _ in range(k):
                    torch.mm(x, w)
    which = {id(r): i % len(batches) for i, r in enumerate(pipe.passes)}
    seen = []

    def on_done(step, r, preds):
        for g, e in zip(preds, eager[which[id(r)]]):
            assert torch.equal(g['pred_boxes'], e['pred_boxes']) and torch.equal(g['pred_scores'], e['pred_scores'])
        seen.append(step)
        load(4)
    load(40)
    assert pipe.run(24, on_done=on_done) == 24 and seen == list(range(24))
    torch.cuda.synchronize()
    for g in pipe.groups:
        assert int(torch.stack([w_.reshape(()) for w_ in g._status_words]).sum()) == 0
    del stop


def test_sampler_failure_is_raised_by_finalize():
    """the host side of the cooperative sampler's failure path: a set error word (here set by hand: the device sets it after
    ~2 s without its partners) reaches finalize() through the pinned copy and raises FpsTimeout once; the word is cleared"""
    from de6d_amd.ops import fused
    from de6d_amd.runtime import load_config, build_model, Det6DGroup, GraphedDet6D
    cfg = load_config('synthetic_models/det6d_65536.yaml')
    model = build_model(cfg, seed=77, device='cuda')
    b, n = 1, 65536
    pts = torch.from_numpy(flat_points(make_batch(8400, b, n))).cuda()
    group = Det6DGroup(model, b, n, 1, torch.cuda.Stream(), points=[pts], main_streams=[torch.cuda.Stream()])
    group.launch()[0].finalize()                                   # a clean run
    group._status_words[0].fill_(1)
    torch.cuda.synchronize()
    r = group.launch()[0]
    with pytest.raises(fused.FpsTimeout):
        r.finalize()
    torch.cuda.synchronize()
    assert int(group._status_words[0]) == 0
    group.launch()[0].finalize()                                   # cleared: the next pass is fine again
    # a group of k = 2 passes: (1) the word sits at the same place for a PARTIAL launch (count < k: fewer scenes through the
    # same workspace), (2) EVERY pass of the failed launch raises, not only the first one finalized
    pts2 = torch.from_numpy(flat_points(make_batch(8401, b, n))).cuda()
    g2 = Det6DGroup(model, b, n, 2, torch.cuda.Stream(), points=[pts, pts2], main_streams=[torch.cuda.Stream(), torch.cuda.Stream()])
    assert fused.fps_status_word(1, n, g2.ws[(0, 0)]).data_ptr() == fused.fps_status_word(2, n, g2.ws[(0, 0)]).data_ptr()
    for r_ in g2.launch():
        r_.finalize()
    g2._status_words[0].fill_(1)
    torch.cuda.synchronize()
    (only,) = g2.launch(count=1)                                   # partial group
    with pytest.raises(fused.FpsTimeout):
        only.finalize()
    torch.cuda.synchronize()
    assert int(g2._status_words[0]) == 0
    g2._status_words[0].fill_(1)
    torch.cuda.synchronize()
    both = g2.launch()
    for r_ in both:                                                # both passes of the failed launch report it
        with pytest.raises(fused.FpsTimeout):
            r_.finalize()
    torch.cuda.synchronize()
    for r_ in g2.launch():                                         # cleared once: the next launch is fine
        r_.finalize()
    # single-graph pass: the word of the sampler captured inside the graph
    runner = GraphedDet6D(model, b, n)
    assert runner._status_words
    runner.launch(pts).finalize()
    # eager: Det6D.forward checks the launches it made
    with torch.no_grad():
        model({'batch_size': b, 'points': pts})
    assert not fused.PENDING_FPS_STATUS
    # C entry: sticky word read and cleared by det6d_fps_fused_status
    ws = fused.fps_workspace(b, n)
    word = fused.fps_status_word(b, n, ws)
    fused.fps_status(b, n, ws)
    word.fill_(1)
    from de6d_amd._lib import Det6dError
    with pytest.raises(Det6dError):
        fused.fps_status(b, n, ws)
    torch.cuda.synchronize()
    assert int(word) == 0


@pytest.mark.parametrize("cfg_name,scene,b,n,shape", [
    ('kitti_models/det6d_car.yaml', 'uniform', 8, 16384, dict(n_main=16, prefetch=4, sampler_streams=6, merge=10)),
    ('kitti_models/det6d_car.yaml', 'beam', 8, 16384, dict(n_main=16, prefetch=4, sampler_streams=6, merge=10)),
    ('synthetic_models/det6d_65536.yaml', 'uniform', 8, 65536, dict(n_main=4, prefetch=2, sampler_streams=6, merge=4)),
])
def test_the_regime_value_times_closed_loop_mid_stream_passes_vs_oracle(oracle_ops, cfg_name, scene, b, n, shape):
    """round-5 review item 5: oracle parity IN the regime bench.py's `value` times, not transitively through the eager model.
    ScenePipeline exactly as the bench builds it for 16384-point scenes (16 main + 6 sampler streams, 4 stages ahead, 80-scene
    passes, closed loop) and for the 65536-point leg (4 main streams, 2 ahead, 32-scene passes), 48 passes in one stream on
    both scene generators; four passes picked from the MIDDLE of the stream — every slot full, ticket-drawn tiles, samplers
    and GEMMs of other passes on the chip — are cloned as they are delivered and compared with oracle/model.py bit for bit."""
    from de6d_amd.runtime import load_config, build_model, ScenePipeline
    from tests.util import beam_batch
    cfg = load_config(cfg_name)
    model = build_model(cfg, seed=1234, device='cuda')
    merge = shape['merge']
    make = beam_batch if scene == 'beam' else make_batch
    n_inputs = 2 if n <= 16384 else 1                       # distinct resident pass inputs the slots cycle through
    batches_np = [flat_points(make(6100 + 30 * j, b, n)) for j in range(n_inputs * merge)]
    batches = [torch.from_numpy(p).cuda() for p in batches_np]
    inputs = ScenePipeline.coalesce(batches, merge)
    assert len(inputs) == n_inputs
    pipe = ScenePipeline(model, b, n, group=1, points=inputs, **shape)
    assert len(pipe.main_streams) == shape['n_main'] and pipe.n_groups == shape['n_main'] + shape['prefetch']
    which = {id(r): i % n_inputs for i, r in enumerate(pipe.passes)}
    n_passes = 48
    picked = (17, 24, 25, 38)                              # passes of the stream (0-based), all with every slot in flight
    got = {}

    def on_done(step, r, preds):
        p = step // merge
        if p in picked:
            # host copies, taken NOW: the slot is relaunched on its own stream as soon as this returns (a device-side clone on
            # the default stream would race with that launch)
            got[step] = (which[id(r)], [{k: v.cpu() for k, v in d.items()} for d in preds])
    assert pipe.run(n_passes * merge, on_done=on_done) == n_passes * merge
    torch.cuda.synchronize()
    assert sorted(got) == [p * merge + j for p in picked for j in range(merge)]
    refs = {}
    for step in sorted(got):
        inp, preds = got[step]
        j = step % merge                                    # batch j of the pass's input
        if (inp, j) not in refs:
            refs[(inp, j)] = oracle_of(cfg, model, batches_np[inp * merge + j], b)['pred_dicts']
        assert len(preds) == b
        for g, w in zip(preds, refs[(inp, j)]):
            np.testing.assert_array_equal(g['pred_boxes'].cpu().numpy(), w['pred_boxes'])
            np.testing.assert_array_equal(g['pred_scores'].cpu().numpy(), w['pred_scores'])
            np.testing.assert_array_equal(g['pred_labels'].cpu().numpy(), w['pred_labels'])
    assert {inp for inp, _ in got.values()} == set(range(n_inputs))
    assert sum(len(p['pred_scores']) for _, ps in got.values() for p in ps) > 0
