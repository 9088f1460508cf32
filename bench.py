#!/usr/bin/env python
"""bench.py — scenes/sec of the Det6D inference hot path on MI355X (BASELINE.json metric).

One "step" = one full pass (PointNet2FSMSG backbone -> PointHeadBox6DVote -> rotated NMS
post-processing, detections sliced per scene) over one batch of 8 synthetic 16384-point
KITTI-like scenes already resident in HBM (BASELINE.json configs[1]).  The passes run through a
two-stage software pipeline (de6d_amd/runtime.py: ScenePipeline / Det6DGroup): stage 1 = pack + the samplers
that depend on nothing but the input cloud (D-FPS of the input, then the d-fps halves of the later layers over
its picks) for a GROUP of 4 passes, one launch each on a sampler stream, issued 4 groups ahead; stage 2 = the
rest of every pass (captured hipGraph segments) on 16 main streams, so the latency-bound sampling rounds (one
workgroup per scene) overlap the MFMA GEMMs of other passes.  Every pass still processes its own batch of 8
scenes and every step is finalised (its per-scene detections materialised on the host side) inside the timed
region.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A STEP is one batch of --batch (8) scenes.  The pipeline coalesces --merge consecutive batches into one PASS (default:
4 batches = 32 scenes; every kernel of the path works scene by scene, so a scene's result does not depend on what
shares its pass, and the larger launches fill the chip better: +12 % over one pass per batch, `--merge 1`).

Timing: the pipeline holds 20 passes (16 in their GEMM stage + 4 whose sampler stage runs ahead), i.e. 80 steps, so a
sync-bracketed run of K steps is mostly pipeline fill + drain when K is small.  `value` is therefore measured
over STEADY-STATE WINDOWS: one continuous stream of steps is issued (a pre-roll that fills the pipeline and lets the
clocks settle, --warmup steps, the windows, a tail that keeps the pipeline full), with barrier + synchronize before
and after the stream.  Every pass leaves a timing event behind its last kernel; step s counts as DELIVERED at the
device time at which every step <= s is complete (passes run on different streams and may finish out of order: this
is what an in-order consumer sees).  A window runs from the delivery of step s to the delivery of step s + K — exactly
K steps are delivered inside it — and `value` is K x batch over the MEAN of the windows starting on consecutive pass
boundaries over >= 768 steps of the stream (a single window of K = 20 steps is 5 passes out of 16 in flight: +-30 %
noisy, and its median is quantised; the mean window is the steady-state time of K steps).  The sync-bracketed time of
K steps on an empty pipeline (`cold`) and the latency of one batch (`latency`) are printed beside it.  After the
timed region every pass's last result is compared with ONE-BATCH eager passes over the same batches (`selfcheck`); a
mismatch exits non-zero.

N > 1: one process per GPU, scenes sharded (weak scaling, 8 scenes per GPU per step), no collective on the data
path; RCCL is used only for the barrier and the max-over-ranks of the window time.  Without torchrun
(`WORLD_SIZE` unset) `--gpus N` starts the N rank processes itself, before anything touches the GPU.  Rank 0
prints ONE JSON line.  N = 1 without `--worker`: an orchestrating parent that never touches the GPU starts the
measuring worker and then the extra legs (dense rows, the other BASELINE configurations, ray-cast LiDAR scenes) as
processes of their own, one at a time, and prints the merged line.
"""
import argparse
import json
import os
import sys
import time

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with the default, the
# passes kept in flight serialise 4-wide.  Measured on MI355X: early in the round (24 streams) 4 queues 2051
# scenes/s, 8 -> 2039, 16 -> 2929, 32 -> 2482; single-graph passes 16 queues / 15 streams 4372, 24 / 22 4549,
# 32 / 24 4434; two-stage pipeline (16 main + 6 sampler streams) 24 queues 9680, 28 -> 7720, 32 -> 8140.
# Must be set before the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from de6d_amd.runtime import ScenePipeline, load_config, build_model, mlp_flops_per_scene, GraphedDet6D  # noqa: E402
from de6d_amd.ops import fused  # noqa: E402
from de6d_amd import synthetic  # noqa: E402

MAIN_STREAMS = []              # the pipeline's main streams (reused by the later legs: fresh streams would come from further
SAMPLER_STREAMS = []           # along PyTorch's stream pool and alias on the hardware queues)
MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0


def synth_points(seed0, b, n, tilt=False, scene='uniform'):
    make = synthetic.beam_batch if scene == 'beam' else synthetic.make_batch
    return synthetic.points_tensor(make(seed0, b, n, tilt=tilt))


def cpu_baseline(cfg, model, pts_np, scenes):
    """the CPU oracle (a port of the reference kernels + the same model math) on `scenes` scenes,
    timed on this host; test infrastructure used only as the reported baseline"""
    from oracle import model as omodel
    from oracle import ops as oops
    oops.build()
    n = pts_np.shape[0] // 8
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    sample = synth_points(5000, scenes, n)
    t0 = time.time()
    omodel.forward(cfg.MODEL, sd, sample, scenes)
    dt = time.time() - t0
    return {"value": scenes / dt, "unit": "scenes/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d scene(s) of the same workload through oracle/model.py (OpenMP on %d threads: GEMM chains over "
                      "rows, FPS over scenes, ball query over centres), %.1f s" % (scenes, os.cpu_count(), dt)}


def index_kernel_rates(model, points, batch, n):
    """pair-evaluations per second of the two search kernels, timed stand-alone with HIP
    events on their launch stream at the workload's SA1 shapes"""
    from de6d_amd.ops import fused as F
    sa = model.backbone_3d.SA_modules[0]
    m = sum(sa.npoint_list)
    rows, xyz = F.pack_points(points, 4)
    xyz = xyz.view(batch, n, 3)
    idx = torch.empty((batch, m), dtype=torch.int32, device='cuda')
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    F.fps_fused(xyz, 0, n, m, None, 1.0, idx, 0)
    e[0].record(); F.fps_fused(xyz, 0, n, m, None, 1.0, idx, 0); e[1].record()
    ctr = F.gather_centres(xyz, idx)
    shells = [(0.0, sa.radii[0], sa.nsamples[0]), (sa.radii[0], sa.radii[1], sa.nsamples[1])]
    F.ball_query_pair(xyz, ctr, shells[0], shells[1])
    e[2].record(); F.ball_query_pair(xyz, ctr, shells[0], shells[1]); e[3].record()
    torch.cuda.synchronize()
    t_fps, t_bq = e[0].elapsed_time(e[1]) * 1e-3, e[2].elapsed_time(e[3]) * 1e-3
    # every sampler of the backbone stand-alone (us per round = per dependent pick): layer, method, points -> picks
    per_round = {}
    cloud = xyz
    for li, sa_l in enumerate(model.backbone_3d.SA_modules):
        n_l = cloud.shape[1]
        idx_l = torch.empty((batch, sum(sa_l.npoint_list)), dtype=torch.int32, device='cuda')
        sc_l = torch.randn((batch, n_l), device='cuda')
        off = 0
        for (lo, hi), method, npoint in zip(sa_l.sample_range_list, sa_l.sample_method_list, sa_l.npoint_list):
            hi = n_l if hi == -1 else hi
            ws_l = F.fps_workspace(batch, hi - lo)
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(2):
                s0.record()
                F.fps_fused(cloud, lo, hi, npoint, sc_l if method == 's-fps' else None, 1.0, idx_l, off, temp=ws_l)
                s1.record()
            torch.cuda.synchronize()
            per_round["SA%d %s %d->%d" % (li + 1, method, hi - lo, npoint)] = round(s0.elapsed_time(s1) * 1e3 / max(npoint - 1, 1), 3)
            off += npoint
        cloud = F.gather_centres(cloud, idx_l)
    out = {"fps_sa1_ms": round(t_fps * 1e3, 3), "fps_pair_evals_per_s": round(batch * (m - 1) * n / t_fps, 0),
           "fps_us_per_round": per_round,
           "ball_query_sa1_ms": round(t_bq * 1e3, 3),
           "bq_pair_evals_per_s_upper_bound_work": round(2.0 * batch * m * n / t_bq, 0)}
    # SURVEY.md 8a rows a15 / a13, stand-alone (not on Det6D's FSMSG path): three_nn + three_interpolate of (B, 64, m)
    # features back onto the n input points (HBM-bound: reads xyz / features, writes (B, 64, n)), rotated NMS of 256 boxes
    from de6d_amd.ops import pointnet2_batch_hip as pn2
    from tests.util import random_boxes
    c = 64
    feats = torch.randn((batch, c, m), device='cuda')
    d2 = torch.empty((batch, n, 3), device='cuda')
    i3 = torch.empty((batch, n, 3), dtype=torch.int32, device='cuda')
    interp = torch.empty((batch, c, n), device='cuda')
    w3 = torch.full((batch, n, 3), 1.0 / 3.0, device='cuda')
    boxes = torch.from_numpy(random_boxes(3, 256)).cuda()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    for rep in range(2):   # second pass is the timed one
        ev[0].record(); pn2.three_nn_wrapper(batch, n, m, xyz, ctr, d2, i3); ev[1].record()
        ev[2].record(); pn2.three_interpolate_wrapper(batch, c, m, n, feats, i3, w3, interp); ev[3].record()
        ev[4].record(); F.nms_device(boxes, 0.1); ev[5].record()
    torch.cuda.synchronize()
    t_nn, t_ip, t_nms = (ev[0].elapsed_time(ev[1]) * 1e-3, ev[2].elapsed_time(ev[3]) * 1e-3, ev[4].elapsed_time(ev[5]) * 1e-3)
    ip_bytes = batch * (c * m * 4 + n * 3 * 8 + c * n * 4)
    out.update({"three_nn_ms": round(t_nn * 1e3, 3), "three_nn_pair_evals_per_s": round(batch * n * m / t_nn, 0),
                "three_interpolate_ms": round(t_ip * 1e3, 3), "three_interpolate_GBps": round(ip_bytes / t_ip / 1e9, 1),
                "nms_256_boxes_us": round(t_nms * 1e6, 1)})
    return out


def input_producer_rate(cfg, batch, n_raw=120000):
    """§8 f1 stage timed stand-alone: B raw KITTI-sized frames already in HBM -> the model's points
    tensor (range mask + sample_points + collate) in one det6d_prepare_points call; HBM roofline on the
    algorithmic bytes (one read of the raw frames + one write of the sampled rows)"""
    from de6d_amd.ops import fused as F
    dc = cfg.DATA_CONFIG
    n_pts = 16384
    for p in dc.DATA_PROCESSOR:
        if p['NAME'] == 'sample_points':
            n_pts = int(p['NUM_POINTS']['test'])
    rng = np.random.default_rng(77)
    r = rng.gamma(2.0, 12.0, batch * n_raw)
    a = rng.uniform(-np.pi, np.pi, batch * n_raw)
    raw = np.stack([r * np.cos(a), r * np.sin(a), rng.normal(-1.2, 0.6, batch * n_raw), rng.uniform(0, 1, batch * n_raw)], 1)
    raw = torch.from_numpy(raw.astype(np.float32)).cuda()
    offsets = torch.arange(0, batch + 1, dtype=torch.int32, device='cuda') * n_raw
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        F.prepare_points(raw, offsets, dc.POINT_CLOUD_RANGE, n_pts, 1)
    reps = 20
    e0.record()
    for _ in range(reps):
        F.prepare_points(raw, offsets, dc.POINT_CLOUD_RANGE, n_pts, 1)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    alg = batch * (n_raw * 16 + n_pts * 20)
    return {"ms_per_batch": round(sec * 1e3, 4), "scenes_per_s": round(batch / sec, 0), "raw_points_per_frame": n_raw,
            "bound": "hbm", "achieved_GBps": round(alg / sec / 1e9, 1), "peak_GBps": 8000.0,
            "frac": round(alg / sec / 8e12, 4)}


def pipeline_rate(cfg, model, batch, n, steps=480, n_raw=120000, group=4, n_main=16, prefetch=4):
    """raw frames -> annotations: det6d_prepare_points (f1) -> captured Det6D passes -> det6d_kitti_annos + one
    D2H + host dictionaries (f2) through the same two-stage pipeline as the headline run; raw frames resident in HBM"""
    from de6d_amd.ops import fused as F
    from de6d_amd.pcdet.datasets import KittiDataset
    from de6d_amd.pcdet.utils.calibration_kitti import Calibration
    dc = cfg.DATA_CONFIG
    rng = np.random.default_rng(78)
    r, a = rng.gamma(2.0, 12.0, batch * n_raw), rng.uniform(-np.pi, np.pi, batch * n_raw)
    raw = np.stack([r * np.cos(a), r * np.sin(a), rng.normal(-1.2, 0.6, batch * n_raw), rng.uniform(0, 1, batch * n_raw)], 1)
    raw = torch.from_numpy(raw.astype(np.float32)).cuda()
    offsets = torch.arange(0, batch + 1, dtype=torch.int32, device='cuda') * n_raw
    calib = Calibration({'P2': np.array([[721.5, 0, 609.6, 44.9], [0, 721.5, 172.9, 0.22], [0, 0, 1, 0.0027]], np.float32),
                         'R0': np.eye(3, dtype=np.float32),
                         'Tr_velo2cam': np.array([[0, -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27]], np.float32)})
    meta = {'calib': [calib] * batch, 'image_shape': np.tile(np.array([[375, 1242]], np.int32), (batch, 1)),
            'frame_id': ['%06d' % i for i in range(batch)]}
    pipe = ScenePipeline(model, batch, n, n_main=n_main, group=group, prefetch=prefetch, sampler_streams=6,
                         main_streams=MAIN_STREAMS[:n_main] or None, samplers=SAMPLER_STREAMS[:6] or None)
    scratch = {}
    for r in pipe.passes:
        scratch[id(r)] = (torch.empty((int(F.L.lib().det6d_prepare_points_workspace_bytes(batch, batch * n_raw)),), dtype=torch.uint8, device='cuda'),
                          torch.empty((batch,), dtype=torch.int32, device='cuda'))
    seed = [0]
    annos = [0]

    def produce(r):
        ws, cnt = scratch[id(r)]
        seed[0] += 1
        F.prepare_points(raw, offsets, dc.POINT_CLOUD_RANGE, n, seed=seed[0], out=r.points, workspace=ws, n_in=cnt)

    def consume(step, r, preds):
        annos[0] += len(KittiDataset.generate_prediction_dicts(meta, preds, cfg.CLASS_NAMES))

    steps = max(8, steps * 8 // batch)
    pipe.run(len(pipe.passes) + 4, feed=produce, on_done=consume)
    torch.cuda.synchronize()
    annos[0] = 0
    t0 = time.perf_counter()
    pipe.run(steps, feed=produce, on_done=consume)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"scenes_per_s": round(annos[0] / dt, 1), "ms_per_pass": round(dt / steps * 1e3, 3), "passes": steps, "scenes_per_pass": batch,
            "timing": "sync-bracketed (fill + drain included)",
            "stages": "raw %d-pt frames (HBM) -> prepare_points -> Det6D passes -> kitti_annos -> annotation dicts (host)" % n_raw}


def family_saturated(replay, n_streams=16, reps=24, streams=None):
    """wall time of the GEMM-family launches of one pass re-issued concurrently on n_streams streams.  Every stream
    writes its own copies of the outputs (and reads its own copies of the intermediates), like passes in flight do;
    weights, point rows and row lists are shared, as in the pipeline."""
    if not replay:
        return None
    # the pipeline's own (now idle) main streams when given: fresh ones would come from further along PyTorch's stream
    # pool and alias on the hardware queues, which serialises the streams that collide
    streams = list(streams)[:n_streams] if streams else [torch.cuda.Stream() for _ in range(n_streams)]
    n_streams = len(streams)
    graphs, keep = [], []
    torch.cuda.synchronize()
    for si, st in enumerate(streams):   # every stream starts at another launch of the pass, as passes in flight do
        own = {}
        for _, out, _ in replay:
            if out.data_ptr() not in own:
                own[out.data_ptr()] = out.clone()
        keep.append(own)

        def ptr_of(t, own=own):
            c = own.get(t.data_ptr())
            return (c if c is not None else t).data_ptr()
        g = torch.cuda.CUDAGraph()
        rot = (si * len(replay)) // n_streams
        with torch.cuda.graph(g, stream=st):
            for issue, _, _ in replay[rot:] + replay[:rot]:
                issue(ptr_of)
        graphs.append(g)

    def run(k):
        for _ in range(k):
            for st, g in zip(streams, graphs):
                with torch.cuda.stream(st):
                    g.replay()
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(reps)
    torch.cuda.synchronize()
    return {"seconds": time.perf_counter() - t0, "passes": n_streams * reps, "streams": n_streams, "replays": reps}


def linear_roofline(model, points, batch, flops_per_scene, streams=None, pmc_tag='uniform'):
    """average achieved TFLOP/s of the dominant kernel family (linear_kernel + the register chain kernels: the
    SA / head MLP GEMMs) measured live with HIP events on the launch stream over one PASS (`batch` scenes: the launches
    the pipeline issues, i.e. --merge batches of 8 scenes per launch).

    The grouped MLPs run on compact row lists (csrc/compact.hip): rows that only repeat another row of the same
    centre (the reference's padding of partly filled balls) are not evaluated.  Three flop counts are reported:
      algorithmic = rows that carry information (sum of min(cnt, nsample) per group; every row of the plain layers),
      issued      = rows the kernels actually multiply (class padding to 4 / 8 / 16 / 32 and 128-row alignment on top),
      dense       = the reference's (centres x nsample) row space, SURVEY.md 8d's 22.583 GFLOP per scene.
    `achieved` prices the ALGORITHMIC flops: padding the kernels add for their own convenience earns nothing."""
    # five eager passes back to back, the MEDIAN duration of every launch (one pass alone, after idle time spent in Python,
    # sometimes runs at a lower clock: 0.90 vs 1.35 ms for the family on the same binary)
    reps, passes = 5, []
    for rep in range(reps):
        fused.LINEAR_EVENTS, fused.LINEAR_REPLAY = [], ([] if rep == reps - 1 else None)
        with torch.no_grad():
            model({'batch_size': batch, 'points': points})
        torch.cuda.synchronize()
        passes.append(fused.LINEAR_EVENTS)
        replay = fused.LINEAR_REPLAY
    fused.LINEAR_EVENTS = fused.LINEAR_REPLAY = None
    saturated = family_saturated(replay, streams=streams)
    ev = passes[-1]
    assert all(len(p) == len(ev) for p in passes)
    dur_ms = [sorted(p[i][0].elapsed_time(p[i][1]) for p in passes)[reps // 2] for i in range(len(ev))]
    total_ms = sum(dur_ms)
    issued = useful = 0.0
    fill, per_launch = [], []
    for (e0, e1, r, k, n), ms in zip(ev, dur_ms):
        us = ms * 1e3
        if torch.is_tensor(r):          # compact list header: [0] issued rows, [7] centres, [8] information rows
            h = r.cpu().tolist()
            issued += 2.0 * h[0] * k * n
            useful += 2.0 * h[8] * k * n
            fill.append((h[7], h[8], h[0]))
            per_launch.append([h[8], k, n, round(us, 1), round(2.0 * h[8] * k * n / us / 1e6, 1)])
        else:
            issued += 2.0 * r * k * n
            useful += 2.0 * r * k * n
            per_launch.append([r, k, n, round(us, 1), round(2.0 * r * k * n / us / 1e6, 1)])
    dense = flops_per_scene * batch
    achieved = useful / (total_ms * 1e-3) / 1e12
    traffic = None
    # committed PMC summaries: profiles/rNN_<tag>_pmc_summary.json with tag "beam" for the ray-cast scenes
    pmc = sorted(f for f in __import__('glob').glob(os.path.join(ROOT, 'profiles', '*pmc_summary.json'))
                 if ('beam' in os.path.basename(f)) == (pmc_tag == 'beam'))
    if os.environ.get('DET6D_DENSE_ROWS'):
        pmc = []
    if pmc:  # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE)
        try:
            traffic = round(json.load(open(pmc[-1]))['_derived']['linear_kernel']['hbm_bytes_per_launch'])
        except Exception:
            traffic = None
    groups = sorted(set(fill))
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic,
            "traffic_source": os.path.basename(pmc[-1]) if pmc and traffic else None,
            "kernel": "linear_kernel<...> + mlp_chain_*_kernel (fp32 MFMA GEMM family, %d launches/pass)" % len(ev),
            "launches_per_pass": len(ev), "avg_launch_us": round(total_ms * 1e3 / max(len(ev), 1), 2),
            "algorithmic_gflop_per_pass": round(useful / 1e9, 2), "issued_gflop_per_pass": round(issued / 1e9, 2),
            "dense_gflop_per_pass": round(dense / 1e9, 2),
            "issued_tflops": round(issued / (total_ms * 1e-3) / 1e12, 2),
            "dense_equivalent_tflops": round(dense / (total_ms * 1e-3) / 1e12, 2),
            "compact_rows_centres_information_issued": groups,
            # every launch of the family in issue order: [information rows, K, N (fused chains: K = 1, N = sum of Cin x Cout), us, TFLOP/s]
            "launches": per_launch,
            # the single longest launch of a pass (mlp_group_kernel of the head's wide radius group) on its own
            "dominant_launch": (lambda x: {"information_rows": x[0], "flop_per_row": 2 * x[1] * x[2], "us": x[3], "tflops": x[4],
                                           "frac": round(x[4] / MFMA_F32_PEAK_TFLOPS, 4),
                                           "share_of_family_time": round(x[3] / (total_ms * 1e3), 3)})(max(per_launch, key=lambda x: x[3])),
            # the same launches with the chip FULL: one pass's GEMM-family launches captured per stream and replayed
            # concurrently on 16 streams, each starting at another launch of the pass and writing its own copies of the
            # outputs, wall clock over 24 replays each.  `achieved` above times the launches one at a time on an idle chip.
            "saturated": None if saturated is None else {
                "tflops": round(useful * saturated["passes"] / saturated["seconds"] / 1e12, 2),
                "frac": round(useful * saturated["passes"] / saturated["seconds"] / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                "issued_tflops": round(issued * saturated["passes"] / saturated["seconds"] / 1e12, 2),
                "streams": saturated["streams"], "replays_per_stream": saturated["replays"],
                "family_ms_per_pass": round(saturated["seconds"] / saturated["passes"] * 1e3, 4)},
            "kernel_ms_per_pass": round(total_ms, 3)}


def child_rate(args, env_extra, extra_args=(), note="", roofline=False):
    """the same bench (same steps / warmup / pipeline shape) in a child process with another switch or workload; the
    parent's GPU state is untouched (the child is a fresh process, nothing is exec'd from this one)"""
    import subprocess
    env = dict(os.environ, **env_extra)
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(args.steps), '--warmup', str(args.warmup),
           '--cpu-scenes', '0', '--no-roofline', '--no-legs', '--worker', '--batch', str(args.batch), '--points', str(args.points), '--cfg', args.cfg,
           '--streams', str(args.streams), '--group', str(args.group), '--prefetch', str(args.prefetch), '--merge', str(args.merge),
           '--sampler-streams', str(args.sampler_streams), '--scene', args.scene] + (['--leg-roofline'] if roofline else []) + list(extra_args)
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res = {"scenes_per_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
               "selfcheck": d.get("selfcheck"), "cold_scenes_per_s": d.get("cold", {}).get("scenes_per_s"),
               "latency_ms_per_batch": d.get("latency", {}).get("ms_per_batch"),
               "latency_under_load_ms": d.get("latency_under_load", {}).get("ms_p50_p99"),
               "window_ms_min_median_max": d.get("config", {}).get("window_ms_min_median_max")}
        if d.get("compact_fill"):
            res["compact_fill"] = {g["group"]: g["fill"] for g in d["compact_fill"]}
        if d.get("roofline"):            # the leg's own GEMM-family roofline (its own flop count: the balls' fill differs)
            res["algorithmic_gflop_per_pass"] = d["roofline"]["algorithmic_gflop_per_pass"]
            res["roofline"] = d["roofline"]
        if note:
            res["note"] = note
        return res
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def compact_fill(model, points, batch):
    """information rows / dense rows of every radius group of one eager pass (how full the balls are: the compact row
    lists evaluate only the information rows, so throughput depends on it)"""
    fused.LINEAR_EVENTS, fused.LINEAR_REPLAY = [], None
    with torch.no_grad():
        model({'batch_size': batch, 'points': points})
    torch.cuda.synchronize()
    ev, fused.LINEAR_EVENTS = fused.LINEAR_EVENTS, None
    names, nsamples = [], []
    for li, sa in enumerate(list(model.backbone_3d.SA_modules) + [model.point_head.SA_module]):
        for gi, ns in enumerate(sa.nsamples):
            names.append("%s-%s" % ("SA%d" % (li + 1) if li < len(model.backbone_3d.SA_modules) else "head", "AB"[gi] if gi < 2 else gi))
            nsamples.append(ns)
    seen, out = set(), []
    for _, _, r, _, _ in ev:                      # the groups' lists appear in model order
        if torch.is_tensor(r) and r.data_ptr() not in seen:
            seen.add(r.data_ptr())
            h = r.cpu().tolist()
            gi = len(out)
            ns = nsamples[gi] if gi < len(nsamples) else None
            out.append({"group": names[gi] if gi < len(names) else str(gi), "centres": h[7], "nsample": ns, "information_rows": h[8],
                        "issued_rows": h[0], "fill": round(h[8] / float(h[7] * ns), 4) if ns and h[7] else None})
    return out


def coalesce_factor(batch, steps, scenes_per_pass=32):
    """batches per pass: the largest d with d * batch <= scenes_per_pass that divides `steps` (a window of K steps is then a
    whole number of passes: exactly K steps are delivered inside it)"""
    return max(d for d in range(1, max(1, scenes_per_pass // max(1, batch)) + 1) if steps % d == 0)


def selfcheck(model, pipe, b):
    """every pass's LAST finalised result against an eager pass over the same batch, bit for bit (the captured segments,
    the grouped first sampler and the stream choreography must not change a single detection)"""
    bad, total = [], 0
    with torch.no_grad():
        for i, r in enumerate(pipe.passes):
            got = r.finalize()
            want = []
            for j in range(len(got) // b):     # a coalesced pass against ONE-BATCH eager passes over its batches
                rows = r.points.shape[0] // (len(got) // b)
                want += model({'batch_size': b, 'points': r.points[j * rows:(j + 1) * rows]})[0]
            torch.cuda.synchronize()
            for sc, (g, w) in enumerate(zip(got, want)):
                total += 1
                same = (g['pred_boxes'].shape == w['pred_boxes'].shape and torch.equal(g['pred_boxes'], w['pred_boxes'])
                        and torch.equal(g['pred_scores'], w['pred_scores']) and torch.equal(g['pred_labels'], w['pred_labels']))
                if not same:
                    bad.append((i, sc))
    return bad, total


def orchestrate(args):
    """N = 1 with the extra legs: this process never touches the GPU.  It starts the measuring worker (headline, roofline,
    CPU baseline), then every leg as a process of its own, ONE AT A TIME, and prints the merged line.  Measured reason: a
    process that merely keeps its 22 idle HIP streams (hardware queues) alive slows a second process on the same GPU by
    25 % (11 255 -> 8 445 scenes/s, scripts/r02/gpu_legtest.sh), so a leg must not run beside its parent's pipeline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ['--worker']
    out = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    if out.returncode != 0 or not lines:
        sys.stdout.write(out.stdout)
        return out.returncode or 1
    line = json.loads(lines[-1])
    if os.environ.get('DET6D_DENSE_ROWS') is None:
        line["dense_rows"] = child_rate(args, {'DET6D_DENSE_ROWS': '1'}, roofline=not args.no_roofline,
                                        note="DET6D_DENSE_ROWS=1: every (centre, nsample slot) row evaluated, as the reference does; "
                                             "the bound for clouds whose every ball is full")
        if args.merge > 1:
            line["one_pass_per_batch"] = child_rate(args, {}, ['--merge', '1', '--group', '4'],
                                                    note="--merge 1 --group 4: every batch of %d scenes is a pass of its own (the shape of the "
                                                         "first half of round 2); `value` coalesces %d batches per pass" % (args.batch, args.merge))
        # the other BASELINE.json configurations (per-GPU share) and ray-cast 64-ring LiDAR scenes, same engine, same
        # steps / warmup / timing; parity of each: tests/test_timed_path_gpu.py, tests/test_model_gpu.py
        legs = [
            ("configs[2] SlopedKITTI Car, batch 8 (sloped scenes, ground-aware pitch branch)",
             ['--cfg', 'slopedkitti_models/det6d_car.yaml', '--tilt']),
            ("configs[3] KITTI 3-class, batch 32 over 8 GPUs = 4 scenes per GPU per step",
             ['--cfg', 'kitti_models/det6d_3class.yaml', '--batch', '4', '--merge', '-1']),
            ("configs[4] 65536 points per scene, batch 64 over 8 GPUs = 8 scenes per GPU per step",
             ['--cfg', 'synthetic_models/det6d_65536.yaml', '--points', '65536', '--batch', '8']),
            ("configs[1] on ray-cast 64-ring LiDAR scenes (range-dependent density: realistic ball fill)", ['--scene', 'beam']),
            ("configs[2] on ray-cast 64-ring LiDAR scenes with a ramp", ['--scene', 'beam', '--tilt', '--cfg', 'slopedkitti_models/det6d_car.yaml']),
        ]
        line["other_configs"] = {name: child_rate(args, {}, extra, roofline=not args.no_roofline and '--scene' in extra)
                                 for name, extra in legs}
        # the reference's timed loop includes load_data_to_gpu (core/tools/eval_utils/eval_utils.py:53-56): the same stream of
        # steps with every batch uploaded from pinned host memory on its pass's sampler stream.  Never `value`.
        line["h2d_inclusive"] = child_rate(args, {}, ['--h2d'],
                                           note="--h2d: every step uploads its %d x %d x 5 floats from pinned host memory (PCIe) before "
                                                "its pass; reported beside `value`, never as `value`" % (args.batch, args.points))
    print(json.dumps(line), flush=True)
    return 0


def spawn_ranks(n):
    """--gpus N without torchrun: start the N rank processes (fresh interpreters, nothing in this one has touched the
    GPU) and return the worst exit code.  Mirrors what core/tools/test.py:137-143 gets from its launcher."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        for p in procs:
            rc = max(rc, abs(p.wait()))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def device_identity():
    pr = torch.cuda.get_device_properties(torch.cuda.current_device())
    ident = getattr(pr, 'uuid', None)
    return "%s|%s" % (torch.cuda.current_device(), ident if ident is not None else getattr(pr, 'pci_bus_id', '?'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=192)
    ap.add_argument('--warmup', type=int, default=48)
    ap.add_argument('--batch', type=int, default=8, help='scenes per GPU per step')
    ap.add_argument('--points', type=int, default=16384)
    ap.add_argument('--merge', type=int, default=-1, help='consecutive batches coalesced into one pass (ScenePipeline merge); default: as many as make a pass of 32 scenes; 1 = one pass per batch')
    ap.add_argument('--streams', type=int, default=-1, help='default 16; main streams = passes in their GEMM stage; main + sampler streams must stay below GPU_MAX_HW_QUEUES (12 -> 8560, 16 -> 9680, 18 -> 7720 scenes/s)')
    ap.add_argument('--prefetch', type=int, default=4, help='groups whose sampler stage is issued ahead of the GEMM stage')
    ap.add_argument('--sampler-streams', type=int, default=6)
    ap.add_argument('--group', type=int, default=-1, help='default 4 (merge 1) / 1 (coalesced passes); passes whose first (input-only) sampler runs as one launch; 0 = every pass is a single captured graph')
    ap.add_argument('--cfg', default='kitti_models/det6d_car.yaml')
    ap.add_argument('--scene', default='uniform', choices=['uniform', 'beam'], help='synthetic scene generator (de6d_amd/synthetic.py)')
    ap.add_argument('--tilt', action='store_true', help='sloped scenes (BASELINE configs[2])')
    ap.add_argument('--distinct-batches', type=int, default=16, help='different resident batches the passes cycle through')
    ap.add_argument('--windows', type=int, default=-1, help='overlapping K-step windows in the stream (default: enough to span 768 steps); their mean is the timed region')
    ap.add_argument('--preroll', type=int, default=-1, help='pre-roll length in pipeline capacities (default 8)')
    ap.add_argument('--cpu-scenes', type=int, default=64, help='scenes timed on the CPU oracle (0 = skip)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--leg-roofline', action='store_true', help='(internal) a child leg that reports its own GEMM-family roofline')
    ap.add_argument('--no-legs', action='store_true', help='skip the child-process legs (dense rows, other BASELINE configs)')
    ap.add_argument('--worker', action='store_true', help='(internal) the measuring process started by the orchestrating parent')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of captured hipGraphs')
    ap.add_argument('--h2d', action='store_true', help='PCIe-inclusive variant: every step uploads its batch from pinned host memory (never the headline value)')
    args = ap.parse_args()
    # pipeline shape: coalesced passes of 32 scenes by default (scripts/r02/gpu_batchsweep.sh, gpu_mergesweep.sh)
    if args.merge < 0:
        # the largest number of batches per pass that keeps a pass <= 32 scenes and divides K (a window of K steps is then a
        # whole number of passes: exactly K steps are delivered inside it)
        args.merge = 1 if (args.no_graph or args.group == 0) else coalesce_factor(args.batch, args.steps)
    if args.group < 0:
        args.group = 4 if args.merge == 1 else 1
    if args.streams < 0:
        args.streams = 16

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))       # nothing above has initialised HIP
    if 'WORLD_SIZE' not in os.environ and not args.worker and not args.no_legs:
        raise SystemExit(orchestrate(args))            # legs run in processes of their own, one at a time (see orchestrate)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if local_rank >= torch.cuda.device_count() and os.environ.get('DET6D_BENCH_BACKEND') != 'gloo':
        raise SystemExit("bench.py: LOCAL_RANK %d but %d visible device(s): one process per GPU" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dist = None
    ranks_seen = [device_identity()]
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('DET6D_BENCH_BACKEND', 'nccl')  # 'nccl' IS RCCL on ROCm; gloo only for dry runs
        dist.init_process_group(backend, rank=rank, world_size=world)
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, device_identity())
        if len(set(ranks_seen)) != world and backend != 'gloo':
            # `--gpus N` on a box with fewer devices would report n_gpus = N from ranks sharing a GPU
            raise SystemExit("bench.py: %d ranks but only %d distinct devices %s (DET6D_BENCH_BACKEND=gloo allows a dry run on "
                             "shared devices)" % (world, len(set(ranks_seen)), sorted(set(ranks_seen))))

    cfg = load_config(args.cfg)
    model = build_model(cfg, seed=1234, device='cuda')
    b, n, merge = args.batch, args.points, args.merge
    # every rank its own scenes; the passes in flight cycle through `distinct_batches` different resident batches
    n_distinct = max(merge, args.distinct_batches)
    batches_np = [synth_points(1000 + (rank * n_distinct + i) * b, b, n, tilt=args.tilt, scene=args.scene) for i in range(n_distinct)]
    batches = [torch.from_numpy(p).cuda() for p in batches_np]
    pts_np, points = batches_np[0], batches[0]
    pass_inputs = ScenePipeline.coalesce(batches, merge)     # resident, one tensor per pass (merge batches back to back)
    depth = max(1, args.streams)
    with torch.no_grad():
        model({'batch_size': b, 'points': points})  # fold weights, load code objects
    torch.cuda.synchronize()

    pipe = None
    if args.no_graph:
        streams = [torch.cuda.Stream() for _ in range(depth)]

        def run(steps, on_done=None):
            inflight, done = [], 0
            with torch.no_grad():
                for i in range(steps):
                    if len(inflight) >= depth:
                        preds = model.finalize(inflight.pop(0))
                        if on_done:
                            on_done(done, None, preds)
                        done += 1
                    with torch.cuda.stream(streams[i % depth]):
                        inflight.append(model.forward_async({'batch_size': b, 'points': batches[i % n_distinct]}))
                for h in inflight:
                    preds = model.finalize(h)
                    if on_done:
                        on_done(done, None, preds)
                    done += 1
            return done
        capacity, k = depth, 1
    elif args.group > 0:
        host_batch = [torch.from_numpy(batches_np[j % n_distinct]).pin_memory() for j in range(merge)] if args.h2d else None
        pipe = ScenePipeline(model, b, n, n_main=depth, group=args.group, prefetch=args.prefetch,
                             sampler_streams=args.sampler_streams, points=None if args.h2d else pass_inputs, merge=merge)
        MAIN_STREAMS.extend(pipe.main_streams)
        SAMPLER_STREAMS.extend(pipe.sampler_streams)

        def run(steps, on_done=None):
            return pipe.run(steps, feed=host_batch, on_done=on_done)
        capacity, k = len(pipe.passes) * merge, pipe.k * merge      # in steps (batches)
    else:
        host_batch = torch.from_numpy(pts_np).pin_memory() if args.h2d else None
        runners = [GraphedDet6D(model, b, n, points=None if args.h2d else batches[i % n_distinct]) for i in range(depth)]
        for r in runners:
            r.launch(host_batch)
        for r in runners:
            r.finalize()

        def run(steps, on_done=None):
            inflight, done = [], 0
            for i in range(steps):
                if len(inflight) >= depth:
                    r0 = inflight.pop(0)
                    preds = r0.finalize()
                    if on_done:
                        on_done(done, r0, preds)
                    done += 1
                inflight.append(runners[i % depth].launch(host_batch))
            for r0 in inflight:
                preds = r0.finalize()
                if on_done:
                    on_done(done, r0, preds)
                done += 1
            return done
        capacity, k = depth, 1

    def bracket():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: steady-state windows of exactly args.steps finalised steps -----------------------------------
    # Passes complete in bursts, so ONE window of a few passes is +-17..30 % noisy (measured).  The stream therefore
    # carries R overlapping windows, each "delivery of step s to delivery of step s + K" with s on consecutive group
    # boundaries, and the MEAN window is the timed region.
    preroll = (args.preroll if args.preroll >= 0 else 8) * capacity   # long enough for clocks / power to settle (~0.2 s)
    preroll += (-(preroll + args.warmup)) % k        # the windows start on group boundaries
    n_windows = args.windows if args.windows > 0 else max(17, -(-768 // k))   # the windows span >= 768 steps of the stream
    n_windows = max(n_windows, -(-2 * capacity // k))    # ... and never less than two pipeline capacities (deliveries are lumpy)
    tail = capacity
    first = preroll + args.warmup - 1
    last = first + (n_windows - 1) * k + args.steps
    stamps, dets, issued = {}, [0], {}
    device_clock = not args.no_graph       # completion times from the device (one timing event per pass) instead of the host

    def on_done(step, r, preds):
        if step <= last:
            stamps[step] = r.stamp if device_clock else time.perf_counter()
            if device_clock:
                issued[step] = r.issued_at
        if first < step <= first + args.steps:
            dets[0] += sum(len(p['pred_scores']) for p in preds)

    bracket()
    GraphedDet6D.host_wait_s = 0.0
    GraphedDet6D.stamp_launches = device_clock
    origin = torch.cuda.Event(enable_timing=True)
    origin.record()
    origin.synchronize()                   # the device is idle here: the event's device time == this host time (to ~10 us)
    t_origin = t_stream = time.perf_counter()
    run(last + 1 + tail, on_done)
    bracket()
    t_stream = time.perf_counter() - t_stream
    GraphedDet6D.stamp_launches = False
    host_wait = GraphedDet6D.host_wait_s
    if device_clock:
        # passes run on different streams and may complete out of step order: step s is DELIVERED when every step <= s is
        # complete (the running maximum of the completion times, what an in-order consumer sees)
        delivered, t_run = {}, 0.0
        for s_ in sorted(stamps):
            t_run = max(t_run, origin.elapsed_time(stamps[s_]) * 1e-3)
            delivered[s_] = t_run
        stamps = delivered
        # per-step latency under load: from the host call that issued the step's pass (stage 1: pack + input-only samplers)
        # to the step's in-order delivery on the device; steps of the timed windows only
        lat = sorted(delivered[s_] - (issued[s_] - t_origin) for s_ in delivered if first < s_ <= last)
        latency_under_load = {"ms_p50_p99": [round(lat[len(lat) // 2] * 1e3, 2), round(lat[min(len(lat) - 1, int(0.99 * len(lat)))] * 1e3, 2)],
                              "ms_min_max": [round(lat[0] * 1e3, 2), round(lat[-1] * 1e3, 2)], "steps": len(lat),
                              "note": "host issue of the step's pass (sampler stage, issued %d group(s) ahead of the GEMM stage) -> "
                                      "in-order delivery of the step on the device clock, with the pipeline full" % args.prefetch}
    else:
        latency_under_load = None
    windows = sorted(stamps[first + j * k + args.steps] - stamps[first + j * k] for j in range(n_windows))
    window_median = windows[len(windows) // 2] if len(windows) % 2 else 0.5 * (windows[len(windows) // 2 - 1] + windows[len(windows) // 2])
    elapsed_own = sum(windows) / len(windows)
    if elapsed_own <= 0.0:
        raise SystemExit("bench.py: the windows cover no time (stream too short for the pipeline): raise --windows or --steps")
    elapsed = elapsed_own

    # ---- cold: K steps on an EMPTY pipeline, synchronize on both sides (fill + drain inside) ----------------------------
    bracket()
    t0 = time.perf_counter()
    run(args.steps)
    bracket()
    cold = time.perf_counter() - t0

    per_rank = [round(args.steps * b / elapsed_own, 1)]
    if dist is not None:
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([elapsed, cold], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, cold = float(t[0].item()), float(t[1].item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, round(args.steps * b / elapsed_own, 1))

    # ---- self-check: what was timed is what the eager model computes ------------------------------------------------------
    check = "skipped"
    if pipe is not None and not args.h2d:
        bad, total = selfcheck(model, pipe, b)
        check = "ok" if not bad else "MISMATCH in (pass, scene) %s" % bad[:8]
        if rank == 0 or bad:
            print("selfcheck rank %d: %d scenes of %d passes compared with the eager model: %s" % (rank, total, len(pipe.passes), check),
                  file=sys.stderr, flush=True)
        if bad:
            raise SystemExit(3)

    if rank == 0:
        flops = mlp_flops_per_scene(model, n)
        knobs = {k_: v for k_, v in os.environ.items() if k_.startswith('DET6D_')}
        line = {
            "metric": "scenes/sec (16384-pt KITTI) at 1/2/4/8 MI355X; 3D mAP parity vs ref",
            "value": round(world * args.steps * b / elapsed, 2),
            "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "KITTI-like Car-only scenes, batch=%d x %d points per GPU per step, Det6D "
                                   "(3-layer FSMSG SA + 6-DoF vote head + rotated NMS), random-init seeded "
                                   "weights; BASELINE.json configs[1]" % (b, n),
                       "cfg": args.cfg, "scenes_per_step_per_gpu": b, "points_per_scene": n, "scene_generator": args.scene,
                       "tilt": args.tilt, "distinct_resident_batches": n_distinct,
                       "timing": "steady-state: one continuous stream of %d steps (%d pre-roll, %d warmup, then the windows, %d tail) "
                                 "between barrier+synchronize; a window = delivery of step s to delivery of step s+%d "
                                 "(exactly %d steps delivered inside; delivered = every step up to it complete on the device); "
                                 "value = mean of %d windows starting on consecutive "
                                 "group boundaries" % (last + 1 + tail, preroll, args.warmup, tail, args.steps, args.steps, n_windows),
                       "preroll_steps": preroll, "tail_steps": tail, "windows": n_windows,
                       "window_ms_min_median_max": [round(windows[0] * 1e3, 3), round(window_median * 1e3, 3), round(windows[-1] * 1e3, 3)],
                       "window_ms_mean": round(elapsed_own * 1e3, 3),
                       "window_clock": "device: completion events of the passes (hipEventElapsedTime)" if device_clock else "host",
                       "batches_per_pass": merge, "scenes_per_pass": b * merge,
                       "pass": "a step is one batch of %d scenes; the pipeline coalesces %d consecutive batches into one pass "
                               "(per-scene results identical to one-batch passes: selfcheck)" % (b, merge),
                       "streams": depth, "sampler_group": args.group, "hipgraph": not args.no_graph,
                       "input": "pinned host, H2D per step" if args.h2d else "resident in HBM",
                       "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "env_knobs": knobs,
                       "parallelism": "scene-sharded x%d, no collective" % world},
            "selfcheck": check,
            "ranks_seen": ranks_seen, "per_rank_scenes_per_s": per_rank,
            "stream_total_s": round(t_stream, 4), "host_blocked_frac": round(host_wait / t_stream, 3),
            "detections_in_window": dets[0],
            "cold": {"scenes_per_s": round(world * args.steps * b / cold, 2), "ms_per_step": round(cold / args.steps * 1e3, 4),
                     "note": "the same %d steps on an empty pipeline, barrier+synchronize on both sides: pipeline fill + drain included" % args.steps},
        }
        if latency_under_load is not None:
            line["latency_under_load"] = latency_under_load
        if world == 1:
            lat = GraphedDet6D(model, b, n, points=points)
            lat.launch(); lat.finalize()
            t0 = time.perf_counter()
            for _ in range(5):
                lat.launch(); lat.finalize()
            line["latency"] = {"ms_per_batch": round((time.perf_counter() - t0) / 5 * 1e3, 3),
                               "note": "one batch of %d scenes, one captured graph on one stream, idle chip" % b}
            del lat
        if world == 1 and not args.no_roofline:
            line["roofline"] = linear_roofline(model, pass_inputs[0], b * merge, flops, streams=MAIN_STREAMS)
            line["roofline"]["scenes_per_pass"] = b * merge
            line["roofline"]["steps_per_pass"] = merge
            line["compact_fill"] = compact_fill(model, points, b)
            line["index_kernels"] = index_kernel_rates(model, points, b, n)
            line["input_producer"] = input_producer_rate(cfg, b)
            pipe = None  # noqa: F841  (frees the captured graphs before the pipeline leg builds its own)
            torch.cuda.empty_cache()
            line["pipeline"] = pipeline_rate(cfg, model, b * merge, n, group=args.group, n_main=depth, prefetch=args.prefetch)
        elif world == 1 and args.leg_roofline:
            line["roofline"] = linear_roofline(model, pass_inputs[0], b * merge, flops, streams=MAIN_STREAMS, pmc_tag=args.scene)
            line["roofline"]["scenes_per_pass"] = b * merge
            line["compact_fill"] = compact_fill(model, points, b)
        elif world == 1:
            line["compact_fill"] = compact_fill(model, points, b)
        if world == 1 and args.cpu_scenes > 0:
            line["cpu_baseline"] = cpu_baseline(cfg, model, pts_np, args.cpu_scenes)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
