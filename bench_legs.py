"""bench_legs.py — the measuring legs of bench.py other than the timed region itself: the GEMM-family roofline (launch by
launch with HIP events + the chip-full replay), the ball-fill report, stand-alone rates of the index kernels and of the input
producer, the raw-frames -> annotations pipeline, the self-check of the timed passes.  bench.py owns the contract (arguments,
the timed region, the JSON line, the orchestration of child legs and ranks) and the CPU-baseline leg (the only place outside
tests/ and smoke() that touches oracle/)."""
import json
import os
import time

import numpy as np
import torch

from de6d_amd.runtime import ScenePipeline
from de6d_amd.ops import fused

ROOT = os.path.dirname(os.path.abspath(__file__))
MAIN_STREAMS = []              # the pipeline's main streams (reused by the later legs: fresh streams would come from further
SAMPLER_STREAMS = []           # along PyTorch's stream pool and alias on the hardware queues)
MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0


_SAMPLER_CODE = r"""
import glob, os, sys, time
out = open(sys.argv[1], 'w')
period = float(sys.argv[2])
parent = os.getppid()
smi = h = None
try:                                   # amdsmi: gfx clock of every XCD + socket power (no HIP context is created)
    import amdsmi as smi
    smi.amdsmi_init()
    h = smi.amdsmi_get_processor_handles()[0]
    smi.amdsmi_get_gpu_metrics_info(h)['current_gfxclks']
except Exception:
    smi = None
hw = [d for d in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*') if glob.glob(d + '/freq1_input')]
while os.getppid() == parent:           # (a bench that died mid-stream must not leave its sampler polling for ever)
    t = time.time()
    clk = pw = None
    try:
        if smi is not None:
            m = smi.amdsmi_get_gpu_metrics_info(h)
            c = [float(x) for x in m['current_gfxclks'] if isinstance(x, (int, float)) and 0 < x < 10000]
            clk = sum(c) / len(c) if c else None
            p = m.get('current_socket_power')
            pw = float(p) if isinstance(p, (int, float)) else None
        elif hw:
            clk = float(open(hw[0] + '/freq1_input').read()) / 1e6
            pw = float(open(hw[0] + '/power1_input').read()) / 1e6
    except Exception:
        pass
    out.write('%.4f %s %s\n' % (t, 'nan' if clk is None else '%.1f' % clk, 'nan' if pw is None else '%.1f' % pw))
    out.flush()
    time.sleep(period)
"""


class ClockPowerSampler(object):
    """round-5 review item 2: average shader clock and socket power OVER THE TIMED STREAM.  A child process that never
    touches HIP polls amdsmi's gpu_metrics (current_gfxclks of the 8 XCDs, current_socket_power; sysfs hwmon freq1_input /
    power1_input as the fall-back) every `period` s and stamps the samples with the host clock; the worker keeps the samples
    between the two barriers of the stream.  The firmware refreshes gpu_metrics about once a millisecond; the values are
    instantaneous, so the mean over a >= 2 s stream is what is reported."""

    def __init__(self, period=0.02):
        import subprocess
        import sys
        import tempfile
        self.path = tempfile.mktemp(prefix='det6d_clk_')
        try:
            self.proc = subprocess.Popen([sys.executable, '-c', _SAMPLER_CODE, self.path, str(period)],
                                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception:
            self.proc = None

    def stop(self, t0, t1):
        """samples with t0 <= t <= t1 (time.time()) -> {'sclk_mhz', 'power_w', ...} or None"""
        if self.proc is None:
            return None
        self.proc.terminate()
        try:
            self.proc.wait(timeout=5)
        except Exception:
            self.proc.kill()
        try:
            rows = [l.split() for l in open(self.path) if len(l.split()) == 3]
            os.remove(self.path)
        except Exception:
            return None
        rows = [(float(a), float(b_), float(c)) for a, b_, c in rows if t0 <= float(a) <= t1]
        clk = sorted(c for _, c, _ in rows if c == c)
        pw = sorted(p for _, _, p in rows if p == p)
        if not clk:
            return None
        return {"sclk_mhz": round(sum(clk) / len(clk), 1), "sclk_mhz_p10_p50_p90": [clk[len(clk) // 10], clk[len(clk) // 2], clk[(9 * len(clk)) // 10]],
                "power_w": round(sum(pw) / len(pw), 1) if pw else None,
                "power_w_p10_p50_p90": [pw[len(pw) // 10], pw[len(pw) // 2], pw[(9 * len(pw)) // 10]] if pw else None,
                "samples": len(clk), "seconds": round(t1 - t0, 3),
                "source": "amdsmi gpu_metrics (current_gfxclks mean over XCDs, current_socket_power; sysfs hwmon fall-back), polled by a "
                          "child process over the timed stream"}


def whole_path_scalars(roof, scenes_per_pass, scenes_per_s, clocks=None, nominal_mhz=2400.0):
    """round-5 review item 6: scalars at the TOP of `roofline` (nested objects do not reach the driver's `parsed`).
    roofline.frac / achieved become the DRIVER-TIMED whole-path figure: algorithmic GFLOP per scene (rows that carry
    information, every pointwise layer) x `value` / the fp32 MFMA peak.  The launch-by-launch family figures move to
    family_frac_idle / family_frac_saturated, the longest launch to dominant_launch_frac."""
    gf = roof["algorithmic_gflop_per_pass"] / float(scenes_per_pass)
    roof["family_frac_idle"] = roof["frac"]
    roof["family_tflops_idle"] = roof["achieved"]
    roof["family_frac_saturated"] = (roof.get("saturated") or {}).get("frac")
    roof["dominant_launch_frac"] = (roof.get("dominant_launch") or {}).get("frac")
    roof["algorithmic_gflop_per_scene"] = round(gf, 4)
    tf = gf * scenes_per_s / 1e3
    roof["achieved"] = round(tf, 2)
    roof["frac"] = round(tf / MFMA_F32_PEAK_TFLOPS, 4)
    roof["whole_path_frac"] = roof["frac"]
    roof["frac_is"] = ("driver-timed whole path: algorithmic_gflop_per_scene x value / peak (every kernel of the pass inside the "
                       "time, only the MLP family's flops counted); family_frac_idle = the GEMM family launch by launch with HIP "
                       "events on an idle chip, family_frac_saturated = the same launches with the chip full of them")
    if clocks:
        roof["sclk_mhz"] = clocks["sclk_mhz"]
        roof["power_w"] = clocks["power_w"]
        roof["whole_path_frac_at_held_clock"] = round(roof["frac"] * nominal_mhz / clocks["sclk_mhz"], 4)
        roof["clock_power_detail"] = clocks
    return roof


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
        for entry in replay:
            for t in [entry[1]] + list(entry[3] if len(entry) > 3 else []):      # outputs + per-replay private tensors (ticket headers)
                if t.data_ptr() not in own:
                    own[t.data_ptr()] = t.clone()
        keep.append(own)

        def ptr_of(t, own=own):
            c = own.get(t.data_ptr())
            return (c if c is not None else t).data_ptr()
        g = torch.cuda.CUDAGraph()
        rot = (si * len(replay)) // n_streams
        with torch.cuda.graph(g, stream=st):
            for entry in replay[rot:] + replay[:rot]:
                entry[0](ptr_of)
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
    # committed PMC summaries: profiles/rNN_<tag>_pmc_summary.json — tag "beam" for the ray-cast scenes, "65536" for the
    # 65536-point scenes, anything else for the benchmark scenes; the latest round's file wins
    def _kind(f):
        f = os.path.basename(f)
        return 'beam' if 'beam' in f else '65536' if '65536' in f else 'uniform'
    def _order(f):        # rNN_<tag> (a round's final collection) after rNNa_<tag> (its first), later rounds last
        m_ = __import__('re').match(r'r(\d+)([a-z]?)_', os.path.basename(f))
        return (int(m_.group(1)), m_.group(2) == '', f) if m_ else (-1, False, f)
    pmc = sorted((f for f in __import__('glob').glob(os.path.join(ROOT, 'profiles', '*pmc_summary.json')) if _kind(f) == pmc_tag), key=_order)
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


#: scenes per coalesced pass the bench aims for, by scene size: 16384-point scenes 80 (round 5: 64-80 scenes per pass deliver 6-7 %
#: more than 32 — 16 290 vs 15 320 scenes/s uniform, 6 960 vs 6 560 ray-cast, 17 900 vs 16 200 sloped — at 2.5x the latency
#: under load: scripts/r05/gpu_t14.sh .. gpu_t16.sh; `operating_points` keeps the smaller shapes in the line), larger scenes 32
#: (65536 points: 1 276 vs 1 285 scenes/s with 80, p50 374 vs 148 ms)
def scenes_per_pass_target(points):
    return 80 if points <= 16384 else 32


def coalesce_factor(batch, steps, scenes_per_pass=80):
    """batches per pass: the largest d with d * batch <= scenes_per_pass that divides `steps` (a window of K steps is then a
    whole number of passes: exactly K steps are delivered inside it)"""
    return max(d for d in range(1, max(1, scenes_per_pass // max(1, batch)) + 1) if steps % d == 0)


def span_windows(capacity, k, requested=-1, floor_steps=768):
    """number of K-step windows (one per pass boundary, k steps apart) in the timed span.  Left to the bench (requested <= 0):
    >= floor_steps steps AND >= 16 pipeline capacities — the passes in flight complete in lock-step bursts, one burst per
    capacity, and the mean of windows over a span of few bursts depends on where its two ends fall inside a burst (rounds 2-4:
    768 steps = 5-10 bursts read ~5 % high; tests/test_host_logic.py reproduces it on a synthetic delivery series).  An explicit
    request is honoured down to two capacities."""
    n = requested if requested > 0 else max(17, -(-floor_steps // k))
    return max(n, -(-(16 if requested <= 0 else 2) * capacity // k))


def window_times(stamps, first, k, steps, n_windows):
    """sorted durations of the windows "delivery of step first + j k -> delivery of step first + j k + steps", j < n_windows"""
    return sorted(stamps[first + j * k + steps] - stamps[first + j * k] for j in range(n_windows))


def delivery_fit(stamps, first, last):
    """least-squares slope [s per step] of delivery time over step index across [first, last]: a rate estimate that does not
    depend on where the span's two ends fall inside a burst of completions"""
    span = [s_ for s_ in range(first, last + 1) if s_ in stamps]
    mean_s, mean_t = sum(span) / len(span), sum(stamps[s_] for s_ in span) / len(span)
    return sum((s_ - mean_s) * (stamps[s_] - mean_t) for s_ in span) / sum((s_ - mean_s) ** 2 for s_ in span)


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


def measured_traffic(bench_path, extra_args=(), passes=3, timeout=420, scenes_per_pass=32):
    """HBM bytes per GEMM-family launch MEASURED in this run (round-4 review: the line used to quote the newest committed PMC
    summary): two child processes of this bench under `rocprofv3 --pmc` — FETCH_SIZE and WRITE_SIZE cannot share a pass
    (MI355X_MICROARCH.md, rocprofv3 PMC slots) — each running `passes` eager passes of `scenes_per_pass` scenes on one stream; bytes =
    FETCH_SIZE x 2 (gfx950 tallies a 128-byte request of a 16-byte-per-lane read at 64 bytes) + WRITE_SIZE, both reported in
    KiB, summed over the family's dispatches of the profiled passes.  Returns None when rocprofv3 is not on the PATH or a pass
    fails (the caller then falls back to the committed summary and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import sys
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None
    fam = ('linear_kernel', 'mlp_chain', 'mlp_group', 'mlp_rows')
    total, launches, n_pass = {}, None, None
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='det6d_pmc_')
        try:
            cmd = ['rocprofv3', '--kernel-trace', '--pmc', ctr, '--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
                   sys.executable, bench_path, '--gpus', '1', '--steps', str(passes + 2), '--warmup', '2', '--batch', str(scenes_per_pass), '--streams', '1',
                   '--no-graph', '--cpu-scenes', '0', '--no-roofline', '--no-legs', '--worker', '--preroll', '0', '--windows', '1'] + list(extra_args)
            env = dict(os.environ, TMPDIR=os.environ.get('TMPDIR', '/tmp'))
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=d)
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            if out.returncode != 0 or not files:
                return None
            acc, disp, packs = 0.0, set(), set()
            for r in csv.DictReader(open(files[0])):
                name = r.get('Kernel_Name', '')
                if 'pack_points_kernel' in name:
                    packs.add(r.get('Dispatch_Id'))
                if r.get('Counter_Name') == ctr and any(k in name for k in fam):
                    acc += float(r['Counter_Value'])
                    disp.add(r.get('Dispatch_Id'))
            total[ctr], launches, n_pass = acc * 1024.0, len(disp), max(1, len(packs))
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if not launches:
        return None
    read, write = 2.0 * total['FETCH_SIZE'], total['WRITE_SIZE']
    return {"hbm_bytes_per_launch": round((read + write) / launches), "hbm_read_bytes_per_pass": round(read / n_pass),
            "hbm_write_bytes_per_pass": round(write / n_pass), "launches_per_pass": round(launches / n_pass, 2), "passes_profiled": n_pass,
            "scenes_per_pass": scenes_per_pass,
            "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two child runs of this bench (eager passes of the timed region's size, one stream); "
                      "FETCH_SIZE x 2 + WRITE_SIZE per MI355X_MICROARCH.md"}

