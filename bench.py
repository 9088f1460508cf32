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

A STEP is one batch of --batch (8) scenes.  The pipeline coalesces --merge consecutive batches into one PASS (default at
K = 20: 10 batches = 80 scenes; every kernel of the path works scene by scene, so a scene's result does not depend on what
shares its pass, and the larger launches fill the chip better: +23 % over one pass per batch, `--merge 1`; +2.6 % over 32-scene
passes at 2.5x their latency under load — `operating_points` keeps the 32-scene shape and the PACED (open-loop) points in the line).

Timing: the pipeline holds 20 passes (16 in their GEMM stage + 4 whose sampler stage runs ahead), i.e. 80 steps, so a
sync-bracketed run of K steps is mostly pipeline fill + drain when K is small.  `value` is therefore measured
over STEADY-STATE WINDOWS: one continuous stream of steps is issued (a pre-roll that fills the pipeline and lets the
clocks settle, --warmup steps, the windows, a tail that keeps the pipeline full), with barrier + synchronize before
and after the stream.  Every pass leaves a timing event behind its last kernel; step s counts as DELIVERED at the
device time at which every step <= s is complete (passes run on different streams and may finish out of order: this
is what an in-order consumer sees).  A window runs from the delivery of step s to the delivery of step s + K — exactly
K steps are delivered inside it — and `value` is K x batch over the MEAN of the windows starting on consecutive pass
boundaries over >= 768 steps and >= 16 pipeline capacities of the stream (a single window of K = 20 steps is 2 passes out
of 16 in flight: +-30 % noisy, and its median is quantised; the mean window is the steady-state time of K steps).  The passes in
flight complete in lock-step bursts, so the mean of windows over a SHORT span depends on where the span's two ends fall inside a
burst: rounds 2-4 spanned 768 steps (~10 bursts) and read ~5 % high (profiles/r05_span_bias_r04_vs_r05.txt).  `crosscheck`
prints two independent rates beside `value`: the least-squares delivery rate over the same span and the whole stream (fill and
drain included) on the host clock.  The sync-bracketed time of
K steps on an empty pipeline (`cold`) and the latency of one batch (`latency`) are printed beside it.  After the
timed region every pass's last result is compared with ONE-BATCH eager passes over the same batches (`selfcheck`); a
mismatch exits non-zero.

N > 1: one process per GPU, scenes sharded (weak scaling, 8 scenes per GPU per step), no collective on the data
path; RCCL is used only for the barrier and the max-over-ranks of the window time.  Without torchrun
(`WORLD_SIZE` unset) `--gpus N` starts the N rank processes itself, before anything touches the GPU.  Rank 0
prints ONE JSON line.  N = 1 without `--worker`: an orchestrating parent that never touches the GPU starts the
measuring worker and then the extra legs (dense rows, the other BASELINE configurations, ray-cast LiDAR scenes) as
processes of their own, one at a time, and prints the merged line.  Every leg that runs on other scenes or rows (dense rows,
ray-cast scenes, 65536-point scenes) reports its own GEMM-family roofline, the 65536-point leg its samplers' us per pick as well
(`fps_us_per_round`); the ray-cast leg's rate and roofline fraction are repeated in `config.raycast_scenes_per_s` /
`roofline.raycast` (both density regimes in the part of the line the driver parses); `latency_under_load` is the per-step
issue -> in-order delivery time with the pipeline full; `latency_b1` ONE frame on an idle chip (the reference's single-frame
callers); `h2d_inclusive` the same stream of steps with every batch uploaded from pinned host memory.

The measuring legs other than the timed region live in bench_legs.py.
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
from de6d_amd import synthetic  # noqa: E402
from bench_legs import (span_windows, window_times, delivery_fit, MAIN_STREAMS, SAMPLER_STREAMS, coalesce_factor, compact_fill, index_kernel_rates,  # noqa: E402
                        input_producer_rate, linear_roofline, measured_traffic, pipeline_rate, scenes_per_pass_target, selfcheck,
                        ClockPowerSampler, whole_path_scalars)


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


def child_rate(args, env_extra, extra_args=(), note="", roofline=False):
    """the same bench (same steps / warmup / pipeline shape) in a child process with another switch or workload; the
    parent's GPU state is untouched (the child is a fresh process, nothing is exec'd from this one)"""
    import subprocess
    env = dict(os.environ, **env_extra)
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(args.steps), '--warmup', str(args.warmup),
           '--cpu-scenes', '0', '--no-roofline', '--no-legs', '--worker', '--batch', str(args.batch), '--points', str(args.points), '--cfg', args.cfg,
           '--streams', str(args.raw_streams), '--group', str(args.group), '--prefetch', str(args.raw_prefetch), '--merge', str(args.merge),
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
        if d.get("index_kernels"):       # stand-alone sampler / query rates of the leg's shapes (us per dependent pick)
            res["fps_us_per_round"] = d["index_kernels"]["fps_us_per_round"]
            res["index_kernels"] = d["index_kernels"]
        if note:
            res["note"] = note
        return res
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


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
             ['--cfg', 'synthetic_models/det6d_65536.yaml', '--points', '65536', '--batch', '8', '--merge', '-1']),
            ("configs[1] on ray-cast 64-ring LiDAR scenes (range-dependent density: realistic ball fill)", ['--scene', 'beam']),
            ("configs[2] on ray-cast 64-ring LiDAR scenes with a ramp", ['--scene', 'beam', '--tilt', '--cfg', 'slopedkitti_models/det6d_car.yaml']),
        ]
        line["other_configs"] = {name: child_rate(args, {}, extra, roofline=not args.no_roofline and ('--scene' in extra or '65536' in extra))
                                 for name, extra in legs}
        # both density regimes in the part of the line the driver parses: `value` is the benchmark generator's (sparse balls),
        # the same engine on ray-cast scenes (KITTI-like density, 2.7x the information rows) is quoted beside it
        ray = line["other_configs"].get(legs[3][0], {})
        if "scenes_per_s" in ray:
            line["config"]["raycast_scenes_per_s"] = ray["scenes_per_s"]
            if isinstance(line.get("roofline"), dict) and ray.get("roofline"):
                line["roofline"]["raycast"] = {"scenes_per_s": ray["scenes_per_s"], "frac": ray["roofline"]["frac"],
                                               "achieved": ray["roofline"]["achieved"],
                                               "family_frac_idle": ray["roofline"].get("family_frac_idle"),
                                               "family_frac_saturated": ray["roofline"].get("family_frac_saturated"),
                                               "sclk_mhz": ray["roofline"].get("sclk_mhz"), "power_w": ray["roofline"].get("power_w"),
                                               "algorithmic_gflop_per_pass": ray["roofline"]["algorithmic_gflop_per_pass"]}
                line["roofline"]["raycast_whole_path_frac"] = ray["roofline"].get("whole_path_frac")
                line["roofline"]["raycast_scenes_per_s"] = ray["scenes_per_s"]
        big = line["other_configs"].get(legs[2][0], {})
        if isinstance(line.get("roofline"), dict) and big.get("roofline"):
            line["roofline"]["cfg5_whole_path_frac"] = big["roofline"].get("whole_path_frac")
            line["roofline"]["cfg5_scenes_per_s"] = big.get("scenes_per_s")
            line["roofline"]["cfg5_sclk_mhz"] = big["roofline"].get("sclk_mhz")
        dense = line.get("dense_rows") or {}
        if isinstance(line.get("roofline"), dict) and dense.get("roofline"):
            line["roofline"]["dense_rows_whole_path_frac"] = dense["roofline"].get("whole_path_frac")
        # Operating points (round-4 review item 7): the same engine at other points of its throughput / latency curve.
        # `value` is the CLOSED loop (a new pass the moment a slot frees: every queue full, latency = queueing).  The PACED legs
        # are an OPEN loop, the shape of a sensor: passes are issued on a fixed cadence (ScenePipeline.run headway) chosen as a
        # fraction of `value`, so nothing queues and the latency is the processing time.  p50 / p99 of host issue -> delivery.
        pts = {}

        def point(name, scenes, streams, ahead, pace):
            m_ = coalesce_factor(args.batch, args.steps, scenes)
            extra = ['--merge', str(m_), '--streams', str(streams), '--prefetch', str(ahead)]
            if pace:
                extra += ['--headway-ms', '%.3f' % (1e3 * m_ * args.batch / (pace * line["value"]))]
            r = child_rate(args, {}, extra)
            pts[name % (m_ * args.batch)] = {k: r.get(k) for k in ("scenes_per_s", "latency_under_load_ms", "selfcheck", "error") if k in r}
            if pace:
                pts[name % (m_ * args.batch)]["headway_ms"] = float(extra[-1])
        point("%d-scene passes, 16 main streams, 4 ahead, closed loop", 32, 16, 4, 0.0)
        point("%d-scene passes, 16 main streams, 2 ahead, paced at 0.93 x value", 32, 16, 2, 0.93)
        point("%d-scene passes, 16 main streams, 2 ahead, paced at 0.97 x value", 80, 16, 2, 0.97)
        pts["%d-scene passes, %d main streams, %d ahead, closed loop (value)" % (args.merge * args.batch, args.streams, args.prefetch)] = {
            "scenes_per_s": line["value"], "latency_under_load_ms": line.get("latency_under_load", {}).get("ms_p50_p99"), "selfcheck": line.get("selfcheck")}
        line["operating_points"] = pts
        line["config"]["operating_points_scenes_per_s_at_p50_ms"] = {
            k: [v.get("scenes_per_s"), (v.get("latency_under_load_ms") or [None])[0]] for k, v in pts.items()}
        # roofline.traffic measured in THIS run (two rocprofv3 --pmc child passes of this bench); the committed summary only as
        # a fallback, named as such
        if isinstance(line.get("roofline"), dict) and not args.no_roofline:
            tr = measured_traffic(os.path.abspath(__file__), ['--cfg', args.cfg, '--points', str(args.points)],
                                  scenes_per_pass=args.batch * args.merge)
            if tr is not None:
                line["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                line["roofline"]["traffic_bytes_per_scene"] = round(tr["hbm_bytes_per_launch"] * tr["launches_per_pass"] / tr["scenes_per_pass"])
                line["roofline"]["traffic_source"] = "measured in this run"
                line["roofline"]["traffic_detail"] = tr
            elif line["roofline"].get("traffic_source"):
                line["roofline"]["traffic_source"] += " (committed summary: rocprofv3 --pmc failed or is missing in this run)"
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
    ap.add_argument('--headway-ms', type=float, default=0.0, help='keep consecutive GEMM-stage launches at least this far apart (ScenePipeline.run headway)')
    ap.add_argument('--dump-deliveries', default='', help='worker: write "step delivery_time_s" of every step of the stream to this file (scripts/r05/delivery_rate.py reads it)')
    ap.add_argument('--merge', type=int, default=-1, help='consecutive batches coalesced into one pass (ScenePipeline merge); default: as many as make a pass of <= 80 scenes (<= 32 for scenes of more than 16384 points) and divide --steps; 1 = one pass per batch')
    ap.add_argument('--streams', type=int, default=-1, help='default 16 (4 for 65536-point scenes: the same throughput at a third of the latency under load); main streams = passes in their GEMM stage; main + sampler streams must stay below GPU_MAX_HW_QUEUES (12 -> 8560, 16 -> 9680, 18 -> 7720 scenes/s)')
    ap.add_argument('--prefetch', type=int, default=-1, help='groups whose sampler stage is issued ahead of the GEMM stage (default 4; 2 for scenes of more than 16384 points)')
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
    # pipeline shape: coalesced passes of up to 80 scenes by default (32 for scenes of more than 16384 points: bench_legs.
    # scenes_per_pass_target; round 2 settled on 32, round 5 re-measured: scripts/r05/gpu_t14.sh .. gpu_t16.sh)
    if args.merge < 0:
        # the largest number of batches per pass that keeps a pass within that size and divides K (a window of K steps is then a
        # whole number of passes: exactly K steps are delivered inside it)
        args.merge = 1 if (args.no_graph or args.group == 0) else coalesce_factor(args.batch, args.steps, scenes_per_pass_target(args.points))
    if args.group < 0:
        args.group = 4 if args.merge == 1 else 1
    # Pipeline depth by scene size: a pass of 65536-point scenes keeps the GEMM family busy 11x longer than one of 16384-point
    # scenes, so 4 passes in flight saturate the chip (1244 scenes/s, p50 under load 154 ms; with 16: 1232 scenes/s, 517 ms:
    # scripts/r04/gpu_tune_65536.sh).  The legs started by child_rate() resolve their own defaults from THEIR --points.
    args.raw_streams, args.raw_prefetch = args.streams, args.prefetch
    if args.streams < 0:
        args.streams = 16 if args.points <= 16384 else max(4, 16 * 16384 // args.points)
    if args.prefetch < 0:
        args.prefetch = 4 if args.points <= 16384 else 2

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # (torch.cuda.device_count() does not initialise the GPU)
        if args.gpus > torch.cuda.device_count() and os.environ.get('DET6D_BENCH_BACKEND') != 'gloo':
            raise SystemExit("bench.py: --gpus %d but %d visible device(s): one process per GPU (DET6D_BENCH_BACKEND=gloo allows a "
                             "dry run on shared devices)" % (args.gpus, torch.cuda.device_count()))
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
    # every rank takes this decision from the same numbers BEFORE the rendezvous: a rank that left alone would keep the others
    # waiting for the store's time-out (10 minutes)
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
    if local_world > torch.cuda.device_count() and os.environ.get('DET6D_BENCH_BACKEND') != 'gloo':
        raise SystemExit("bench.py: %d ranks on this node but %d visible device(s): one process per GPU "
                         "(DET6D_BENCH_BACKEND=gloo allows a dry run on shared devices)" % (local_world, torch.cuda.device_count()))
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
            return pipe.run(steps, feed=host_batch, on_done=on_done, headway=args.headway_ms * 1e-3)
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
    # the windows span >= 768 steps and >= 16 pipeline capacities of the stream (bench_legs.span_windows: why 16)
    n_windows = span_windows(capacity, k, args.windows)
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

    sampler = ClockPowerSampler() if (rank == 0 and world == 1) else None     # shader clock + socket power over the stream
    bracket()
    GraphedDet6D.host_wait_s = 0.0
    GraphedDet6D.stamp_launches = device_clock
    origin = torch.cuda.Event(enable_timing=True)
    origin.record()
    origin.synchronize()                   # the device is idle here: the event's device time == this host time (to ~10 us)
    t_origin = t_stream = time.perf_counter()
    t_wall0 = time.time()
    run(last + 1 + tail, on_done)
    bracket()
    t_stream = time.perf_counter() - t_stream
    # (the pipeline's fill and the pre-roll are left out: the first quarter of the stream)
    clocks = sampler.stop(t_wall0 + 0.25 * t_stream, t_wall0 + t_stream) if sampler is not None else None
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
    windows = window_times(stamps, first, k, args.steps, n_windows)
    window_median = windows[len(windows) // 2] if len(windows) % 2 else 0.5 * (windows[len(windows) // 2 - 1] + windows[len(windows) // 2])
    elapsed_own = sum(windows) / len(windows)
    if elapsed_own <= 0.0:
        raise SystemExit("bench.py: the windows cover no time (stream too short for the pipeline): raise --windows or --steps")
    elapsed = elapsed_own
    # two cross-checks of the window mean, printed beside it (neither is `value`): the least-squares slope of delivery time over
    # step index across the same span (insensitive to where the span's two ends fall inside a burst of completions), and the
    # whole stream on the host clock (pipeline fill, drain and the pre-roll included: a lower bound)
    slope = delivery_fit(stamps, first, last)
    if args.dump_deliveries:
        with open(args.dump_deliveries, 'w') as f:      # step index, in-order delivery time [s] since the stream started
            f.write(''.join('%d %.6f\n' % (s_, stamps[s_]) for s_ in sorted(stamps)))
    crosscheck = {"fit_scenes_per_s": round(world * b / slope, 1), "whole_stream_scenes_per_s": round(world * (last + 1 + tail) * b / t_stream, 1),
                  "note": "fit = least-squares slope of in-order delivery time over step index across the windows' span; whole stream = "
                          "all %d steps over the host clock between the two barriers (fill + drain inside); per-rank figures x ranks" % (last + 1 + tail)}

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
                       "span_note": "from round 5 on the windows span >= 16 pipeline capacities; rounds 2-4 spanned 768 steps (~10 lock-step "
                                    "completion bursts) and read ~5 % high: the round-4 build gives 14 544 scenes/s on that span and 13 806 "
                                    "on this one (profiles/r05_span_bias_r04_vs_r05.txt); `crosscheck` carries two independent rates",
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
            "crosscheck": crosscheck, "stream_total_s": round(t_stream, 4), "host_blocked_frac": round(host_wait / t_stream, 3),
            "detections_in_window": dets[0],
            "cold": {"scenes_per_s": round(world * args.steps * b / cold, 2), "ms_per_step": round(cold / args.steps * 1e3, 4),
                     "note": "the same %d steps on an empty pipeline, barrier+synchronize on both sides: pipeline fill + drain included" % args.steps},
        }
        if latency_under_load is not None:
            line["latency_under_load"] = latency_under_load
        if world == 1:
            # Both runners are built BEFORE either is timed and freed after both: releasing a captured graph (its memory pool,
            # its queues) right before a measurement inflates it (scripts/r04/latency_b1_ctx.py: 3.35 -> 3.7 ms after `del`
            # of one runner, 13 ms for the first frame after a pipeline was freed).  Medians over the timed launches.
            lat = GraphedDet6D(model, b, n, points=points)
            # one frame, idle chip: the shape of the reference's single-frame callers (sim/gazebo/src/detection/script/
            # detection.py:108-126,185-188; core/tools/demo.py)
            one = torch.from_numpy(synth_points(4242, 1, n, tilt=args.tilt, scene=args.scene)).cuda()
            lat1 = GraphedDet6D(model, 1, n, points=one)
            torch.cuda.synchronize()

            def _median_ms(runner, reps):
                for _ in range(2):
                    runner.launch(); runner.finalize()
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    runner.launch(); runner.finalize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                ts.sort()
                return round(ts[len(ts) // 2], 3), round(ts[0], 3), round(ts[-1], 3)
            med, lo, hi = _median_ms(lat, 7)
            line["latency"] = {"ms_per_batch": med, "ms_min_max": [lo, hi],
                               "note": "one batch of %d scenes, one captured graph on one stream, idle chip; median of 7" % b}
            med, lo, hi = _median_ms(lat1, 11)
            line["latency_b1"] = {"ms_per_frame": med, "ms_min_max": [lo, hi],
                                  "note": "ONE scene of %d points, one captured graph on one stream, idle chip, host launch -> "
                                          "detections on the host; median of 11" % n}
            line["config"]["latency_b1_ms_per_frame"] = line["latency_b1"]["ms_per_frame"]
            line["config"]["latency_ms_per_batch"] = line["latency"]["ms_per_batch"]
            del lat, lat1, one
        if world == 1 and not args.no_roofline:
            line["roofline"] = linear_roofline(model, pass_inputs[0], b * merge, flops, streams=MAIN_STREAMS)
            line["roofline"]["scenes_per_pass"] = b * merge
            line["roofline"]["steps_per_pass"] = merge
            whole_path_scalars(line["roofline"], b * merge, line["value"], clocks)
            line["compact_fill"] = compact_fill(model, points, b)
            line["index_kernels"] = index_kernel_rates(model, points, b, n)
            line["input_producer"] = input_producer_rate(cfg, b)
            pipe = None  # noqa: F841  (frees the captured graphs before the pipeline leg builds its own)
            torch.cuda.empty_cache()
            line["pipeline"] = pipeline_rate(cfg, model, b * merge, n, group=args.group, n_main=depth, prefetch=args.prefetch)
        elif world == 1 and args.leg_roofline:
            line["roofline"] = linear_roofline(model, pass_inputs[0], b * merge, flops, streams=MAIN_STREAMS,
                                               pmc_tag='65536' if n == 65536 else args.scene)
            line["roofline"]["scenes_per_pass"] = b * merge
            whole_path_scalars(line["roofline"], b * merge, line["value"], clocks)
            line["compact_fill"] = compact_fill(model, points, b)
            if n != 16384:               # the samplers / queries of this leg's shapes (the 65536-point cooperative sampler)
                line["index_kernels"] = index_kernel_rates(model, points, b, n)
        elif world == 1:
            line["compact_fill"] = compact_fill(model, points, b)
        if clocks:
            line["clocks"] = {"sclk_mhz": clocks["sclk_mhz"], "power_w": clocks["power_w"]}
        if world == 1 and args.cpu_scenes > 0:
            line["cpu_baseline"] = cpu_baseline(cfg, model, pts_np, args.cpu_scenes)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
