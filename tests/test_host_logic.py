"""Host-side mirror of the reference API (no GPU): config, registries, state-dict layout,
BN folding, workload accounting."""
import json
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_yaml_and_overrides(tmp_path):
    from de6d_amd.pcdet.config import EasyDict, cfg_from_list, cfg_from_yaml_file, merge_new_config
    cfg = cfg_from_yaml_file(os.path.join(ROOT, 'de6d_amd/cfgs/kitti_models/det6d_car.yaml'), EasyDict())
    assert cfg.MODEL.NAME == 'Det6D' and cfg.MODEL.BACKBONE_3D.SA_CONFIG.NPOINT_LIST[1] == [512, 512]
    assert cfg.DATA_CONFIG.DATA_PROCESSOR[1].NUM_POINTS.test == 16384
    assert cfg.MODEL.POINT_HEAD.get('NOT_THERE', 7) == 7
    cfg_from_list(['MODEL.POST_PROCESSING.SCORE_THRESH', '0.3', 'MODEL.POST_PROCESSING.NMS_CONFIG.NMS_TYPE', 'nms_gpu',
                   'MODEL.POINT_HEAD.SAMPLE_RANGE', '0,128'], cfg)
    assert cfg.MODEL.POST_PROCESSING.SCORE_THRESH == 0.3 and cfg.MODEL.POINT_HEAD.SAMPLE_RANGE == [0, 128]
    with pytest.raises(AssertionError):
        cfg_from_list(['MODEL.NO_SUCH_KEY', '1'], cfg)
    base = tmp_path / 'base.yaml'
    base.write_text("A: {x: 1, y: 2}\nB: 3\n")
    child = tmp_path / 'child.yaml'
    child.write_text("_BASE_CONFIG_: %s\nA: {y: 5}\n" % base)
    merged = cfg_from_yaml_file(str(child), EasyDict())
    assert merged.A.x == 1 and merged.A.y == 5 and merged.B == 3
    assert merge_new_config(EasyDict(), {'K': {'z': [1, {'q': 2}]}}).K.z[1].q == 2


def test_state_dict_layout_equals_the_reference():
    """key names, order and shapes recorded from the reference's own build_network
    (tests/golden/det6d_car_state_dict.json) -> reference checkpoints load unchanged"""
    from de6d_amd.runtime import load_config, build_model
    ref = json.load(open(os.path.join(ROOT, 'tests/golden/det6d_car_state_dict.json')))
    model = build_model(load_config('kitti_models/det6d_car.yaml'))
    sd = model.state_dict()
    assert list(sd.keys()) == ref['keys']
    assert [list(v.shape) for v in sd.values()] == ref['shapes']
    assert sum(p.numel() for p in model.parameters()) == ref['n_params'] == 2356422


def test_registries_and_checkpoint_roundtrip(tmp_path):
    from de6d_amd.pcdet.models import backbones_3d, dense_heads, detectors
    from de6d_amd.runtime import load_config, build_model
    assert 'PointNet2FSMSG' in backbones_3d.__all__ and 'PointHeadBox6DVote' in dense_heads.__all__
    assert 'Det6D' in detectors.__all__
    cfg = load_config('synthetic_models/det6d_tiny.yaml')
    a = build_model(cfg, seed=1)
    path = tmp_path / 'checkpoint_epoch_80.pth'
    torch.save({'epoch': 80, 'it': 1, 'model_state': a.state_dict(), 'version': 'pcdet+0.5.2'}, str(path))

    class Log:
        def info(self, *_):
            pass
    b = build_model(cfg, seed=2)
    b.load_params_from_file(str(path), Log(), to_cpu=True)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    with pytest.raises(NotImplementedError):
        bad = load_config('synthetic_models/det6d_tiny.yaml')
        bad.MODEL['ROI_HEAD'] = {'NAME': 'x'}
        build_model(bad)


def test_bn_folding_matches_torch_eval():
    from de6d_amd.pcdet.ops.pointnet2.pointnet2_batch.pointnet2_modules import fold_sequential
    torch.manual_seed(0)
    seq = torch.nn.Sequential(torch.nn.Conv2d(7, 10, 1, bias=False), torch.nn.BatchNorm2d(10), torch.nn.ReLU(),
                              torch.nn.Conv2d(10, 5, 1, bias=False), torch.nn.BatchNorm2d(5), torch.nn.ReLU())
    for m in seq:
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(); m.running_var.uniform_(0.5, 1.5); m.weight.data.normal_(); m.bias.data.normal_()
    seq.eval()
    x = torch.randn(3, 7, 11, 4)
    want = seq(x).permute(0, 2, 3, 1).reshape(-1, 5).detach().numpy()
    layers = fold_sequential(seq, 8)
    h = np.zeros((3 * 11 * 4, 8), np.float32)
    h[:, :7] = x.permute(0, 2, 3, 1).reshape(-1, 7).numpy()
    for w, s, cout, act in layers:
        h = h @ w
        h[:, :cout] += s
        if act:
            h = np.maximum(h, 0)
    np.testing.assert_allclose(h[:, :5], want, rtol=1e-4, atol=1e-5)
    assert layers[0][0].shape == (8, 12) and layers[1][0].shape == (12, 8)


def test_algorithmic_flops_match_survey():
    from de6d_amd.runtime import load_config, build_model, mlp_flops_per_scene
    model = build_model(load_config('kitti_models/det6d_car.yaml'))
    assert abs(mlp_flops_per_scene(model, 16384) / 1e9 - 22.583) < 0.01


def test_training_mode_is_refused():
    from de6d_amd.runtime import load_config, build_model
    model = build_model(load_config('synthetic_models/det6d_tiny.yaml'))
    model.train()
    with pytest.raises(NotImplementedError):
        model({'batch_size': 1, 'points': torch.zeros((2048, 5))})


def test_any_load_state_dict_drops_the_folded_weights():
    """a plain nn.Module.load_state_dict() must invalidate the folded Conv+BN matrices and bump the version
    captured graphs check"""
    from de6d_amd.runtime import load_config, build_model
    model = build_model(load_config('synthetic_models/det6d_tiny.yaml'), seed=3)
    sa = model.backbone_3d.SA_modules[0]
    sa._folded = {'device': 'stale'}
    model.point_head._folded = {'device': 'stale'}
    v0 = model.weights_version
    model.load_state_dict(model.state_dict())          # NOT the private _load_state_dict
    assert sa._folded is None and model.point_head._folded is None
    assert model.weights_version > v0
    v1 = model.weights_version
    model.eval()                                       # no mode change: nothing to drop
    assert model.weights_version == v1


def test_hoist_plan_is_the_input_only_sampler_chain():
    """runtime.hoist_plan: Det6D's d-fps chain (4096 of the input -> 512 of those -> 256 of those) depends on nothing but
    the input cloud; the s-fps halves (confidence-weighted) do not"""
    from de6d_amd.runtime import load_config, build_model, hoist_plan
    model = build_model(load_config('kitti_models/det6d_car.yaml'), seed=1)
    plan = hoist_plan(list(model.backbone_3d.SA_modules), 16384)
    assert [(s['layer'], s['j'], s['src'], s['lo'], s['hi'], s['m'], s['offset'], s['bias'], s['feeds']) for s in plan] == [
        (0, 0, None, 0, 16384, 4096, 0, 0, True),
        (1, 1, (0, 0), 0, 4096, 512, 512, 0, True),
        (2, 1, (1, 1), 0, 512, 256, 256, 512, False)]


def test_ray_cast_scene_generator():
    """de6d_amd/synthetic.py: beam_scene is deterministic, returns exactly n in-range points through the reference's
    near / far sampling rule, and its density falls with range like a spinning LiDAR's (what makes the ball fill realistic)"""
    from de6d_amd.synthetic import beam_scene, sample_points_rule, points_tensor, beam_batch
    a, b = beam_scene(7, 16384), beam_scene(7, 16384)
    np.testing.assert_array_equal(a, b)
    assert a.shape == (16384, 4) and a.dtype == np.float32 and np.isfinite(a).all()
    assert a[:, 0].min() >= 0 and a[:, 0].max() <= 70.4 and np.abs(a[:, 1]).max() <= 40 and (a[:, 3] >= 0).all() and (a[:, 3] <= 1).all()
    d = np.linalg.norm(a[:, :3], axis=1)
    near, mid = (d < 15).sum(), ((d >= 15) & (d < 30)).sum()
    assert near > 2 * mid > 0                                     # range-dependent density
    assert not np.array_equal(a, beam_scene(8, 16384))
    # the sampling rule: far points (>= 40 m) all survive a down-sampling, short clouds are padded with duplicates
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.uniform(0, 20, (5000, 4)), np.concatenate([rng.uniform(45, 60, (300, 1)), rng.uniform(-5, 5, (300, 3))], 1)]).astype(np.float32)
    out = sample_points_rule(pts, 2000, np.random.default_rng(1))
    assert out.shape == (2000, 4) and (np.linalg.norm(out[:, :3], axis=1) >= 40).sum() == 300
    short = sample_points_rule(pts[:100], 256, np.random.default_rng(2))
    assert short.shape == (256, 4) and len(np.unique(short, axis=0)) == 100
    flat = points_tensor(beam_batch(3, 2, 2048))
    assert flat.shape == (4096, 5) and (flat[:2048, 0] == 0).all() and (flat[2048:, 0] == 1).all()


def test_scene_pipeline_step_accounting_with_coalesced_passes():
    """ScenePipeline.run counts BATCHES (steps) while its passes hold `merge` batches each: launch order, prefetch distance,
    on_done once per batch in step order with that batch's slice of the pass, and a stream whose length is not a multiple
    of `merge` ends with a full pass that reports its leading batches only.  (Fake groups: the choreography is host logic.)"""
    from de6d_amd.runtime import ScenePipeline

    log = []

    class FakePass(object):
        def __init__(self, name, scenes):
            self.name, self.scenes, self.launched = name, scenes, 0

        def finalize(self):
            return [(self.name, self.launched, i) for i in range(self.scenes)]

    class FakeGroup(object):
        def __init__(self, g, k, scenes):
            self.runners = [FakePass((g, j), scenes) for j in range(k)]
            self.count = k

        def launch_front(self, feed, count):
            self.count = count
            log.append(('front', self.runners[0].name[0], count))

        def launch_rest(self):
            log.append(('rest', self.runners[0].name[0]))
            for r in self.runners[:self.count]:
                r.launched += 1
            return self.runners[:self.count]

    for merge, k, n_groups, prefetch, steps in ((4, 1, 6, 2, 21), (2, 2, 4, 2, 13), (1, 4, 3, 1, 9)):
        pipe = object.__new__(ScenePipeline)
        pipe.merge, pipe.step_scenes, pipe.k, pipe.prefetch, pipe.n_groups = merge, 8, k, prefetch, n_groups
        pipe.groups = [FakeGroup(g, k, 8 * merge) for g in range(n_groups)]
        del log[:]
        seen = []

        def on_done(step, r, preds):
            seen.append((step, r.name, [p[2] for p in preds]))
        assert pipe.run(steps, on_done=on_done) == steps
        assert [s for s, _, _ in seen] == list(range(steps))
        n_pass = -(-steps // merge)
        for step, name, scenes in seen:
            p = step // merge                                   # pass in launch order
            assert name == ((p // k) % n_groups, p % k)
            assert scenes == list(range((step % merge) * 8, (step % merge + 1) * 8))      # that batch's slice of the pass
        fronts = [e for e in log if e[0] == 'front']
        rests = [e for e in log if e[0] == 'rest']
        assert sum(c for _, _, c in fronts) == n_pass and len(rests) == len(fronts)
        # a group's sampler stage is issued `prefetch` groups ahead of its GEMM stage
        first_rest = next(i for i, e in enumerate(log) if e[0] == 'rest')
        assert sum(1 for e in log[:first_rest] if e[0] == 'front') == min(prefetch + 1, len(fronts))

    batches = [torch.full((6, 5), float(i)) for i in range(5)]
    merged = ScenePipeline.coalesce(batches, 2)
    assert len(merged) == 3 and all(m.shape == (12, 5) for m in merged)
    assert merged[2][:6].eq(4).all() and merged[2][6:].eq(0).all()      # cyclic
    assert ScenePipeline.coalesce(batches, 1)[3] is batches[3]


def test_bench_coalesce_factor():
    """bench.py: batches per pass = largest divisor of K that keeps a pass within the target size (80 scenes for 16384-point
    scenes, 32 for larger ones)"""
    import bench
    from bench_legs import scenes_per_pass_target
    assert scenes_per_pass_target(16384) == 80 and scenes_per_pass_target(65536) == 32
    assert bench.coalesce_factor(8, 20, 32) == 4 and bench.coalesce_factor(8, 192, 32) == 4
    assert bench.coalesce_factor(8, 7, 32) == 1 and bench.coalesce_factor(8, 6, 32) == 3
    assert bench.coalesce_factor(4, 20, 32) == 5 and bench.coalesce_factor(4, 192, 32) == 8
    assert bench.coalesce_factor(32, 20, 32) == 1 and bench.coalesce_factor(64, 20, 32) == 1
    assert bench.coalesce_factor(8, 20) == 10 and bench.coalesce_factor(8, 192) == 8 and bench.coalesce_factor(8, 8) == 8
    assert bench.coalesce_factor(4, 20) == 20 and bench.coalesce_factor(8, 7) == 7


def test_timed_span_is_long_enough_for_lock_step_deliveries():
    """bench.py's `value` is K x batch over the MEAN of K-step windows starting on consecutive pass boundaries; that mean
    telescopes to (delivery of the span's last passes - delivery of its first) / passes.  With the passes in flight completing
    in lock-step (16 at the same instant, then nothing for a burst period) a span of few bursts that starts right after a
    burst — as the pre-roll of a whole number of capacities makes it — reads high: rounds 2-4 spanned 768 steps and quoted
    ~5 % too much (profiles/r05_span_bias_r04_vs_r05.txt).  bench_legs.span_windows now spans >= 16 capacities, and the
    least-squares fit printed beside it does not care where the ends fall."""
    from bench_legs import span_windows, window_times, delivery_fit
    merge, in_flight, ahead, K = 10, 16, 4, 20              # 80-scene passes: 16 in the GEMM stage + 4 sampler stages ahead
    capacity, k = (in_flight + ahead) * merge, merge
    period = 5.39e-3                                         # true seconds per pass (14 850 scenes/s at 80 scenes per pass)
    burst = in_flight * period

    def stamps_of(n_steps):                                  # every pass of a burst is delivered when the burst completes
        return {s: (s // merge // in_flight + 1) * burst for s in range(n_steps)}

    def value(n_windows, first):
        stamps = stamps_of(first + n_windows * k + K + capacity)
        w = window_times(stamps, first, k, K, n_windows)
        return K / (sum(w) / len(w)), 1.0 / delivery_fit(stamps, first, first + (n_windows - 1) * k + K)
    true = 1.0 / (period / merge)                            # steps per second
    preroll = 8 * capacity + (-(8 * capacity + 5)) % k       # bench.py: pre-roll of 8 capacities, the windows start on a pass boundary
    first = preroll + 5 - 1                                  # ... after 5 warmup steps
    short, short_fit = value(max(17, -(-768 // k)), first)   # rounds 2-4: 77 windows = 4.8 bursts
    assert short / true > 1.10                               # ... reads > 10 % high on ideal lock-step (measured on the chip: ~5-14 %)
    assert abs(short_fit / true - 1.0) < 0.06                # the fit over the same short span is already within a few per cent
    n = span_windows(capacity, k)
    assert n * k >= 16 * capacity and n >= -(-768 // k)
    long_, long_fit = value(n, first)
    assert abs(long_ / true - 1.0) < 0.045 and abs(long_fit / true - 1.0) < 0.01
    assert span_windows(capacity, k, requested=3) == -(-2 * capacity // k)      # an explicit request: at least two capacities
    assert span_windows(20, 1) == 768                        # one pass per batch: 768 steps are already 38 capacities


def test_mlp_rows_supported_query_mirrors_the_launch_checks():
    """det6d_mlp_rows_supported (host logic of csrc/mlp_rows.hip: chain structure, widths, the 160 KB of LDS a 32-row tile's
    two activation buffers may take) decides fused.mlp_rows_eligible: a stack that does not fit is routed through one
    det6d_linear per layer instead of failing inside the forward pass"""
    import torch
    from de6d_amd.ops import fused
    z = lambda *s: torch.zeros(s)   # noqa: E731
    out = z(4, 1024)
    ok = [[(z(256, 128), 0, None, 256, 128, 1, None, 0), (z(128, 64), 0, None, 128, 64, 1, None, 0), (z(64, 4), 0, None, 64, 1, 0, z(4, 4), 0)]]
    assert fused.mlp_rows_eligible(256, ok)
    towers = [[(z(512, 128), 0, None, 512, 128, 1, None, 0), (z(128, 4), 0, None, 128, 1, 0, z(4, 4), 0)],
              [(z(512, 128), 0, None, 512, 128, 1, None, 0), (z(128, 32), 0, None, 128, 32, 0, z(4, 32), 0)]]
    assert fused.mlp_rows_eligible(512, towers)                       # K-chunked wide input
    assert not fused.mlp_rows_eligible(1024, [[(z(1024, 1024), 0, None, 1024, 1024, 1, None, 0), (z(1024, 1024), 0, None, 1024, 1024, 1, out, 0)]])
    assert not fused.mlp_rows_eligible(96, [[(z(96, 50), 0, None, 96, 50, 1, None, 0), (z(50, 4), 0, None, 50, 1, 0, z(4, 4), 0)]])   # hidden width % 32
    assert not fused.mlp_rows_eligible(256, [[(z(256, 128), 0, None, 256, 128, 1, None, 0)]])    # last layer without an output
    assert not fused.mlp_rows_eligible(256, ok + ok + ok)                                           # three chains


def test_bench_refuses_more_ranks_than_devices():
    """`bench.py --gpus N` with fewer than N visible devices exits non-zero BEFORE any rank is started or any rendezvous is
    attempted (a rank that left alone would keep the others waiting for the store's ten-minute time-out); decided from
    torch.cuda.device_count(), which does not initialise the GPU.  Here: no device at all."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a machine with fewer than two devices")
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'DET6D_BENCH_BACKEND')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1'], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert 'visible device' in (out.stderr + out.stdout)


def test_hardware_queue_rule(monkeypatch):
    """GPU_MAX_HW_QUEUES is read when HIP initialises: importing the package exports 24 while nothing has touched the GPU;
    require_hw_queues raises (instead of the round-4 warning) when fewer queues than streams are in effect, unless the caller
    asks for the aliased form"""
    import subprocess
    import sys
    import de6d_amd
    from de6d_amd import runtime
    # a fresh interpreter that has not chosen a value gets 24 from the import, one that has keeps its own
    code = "import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import de6d_amd; print(os.environ['GPU_MAX_HW_QUEUES'], de6d_amd.HW_QUEUES_AT_IMPORT)"
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.stdout.split() == ['24', 'None'], out.stdout + out.stderr
    code = "import os; os.environ['GPU_MAX_HW_QUEUES'] = '8'; import de6d_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.stdout.split() == ['8']
    monkeypatch.setattr(de6d_amd, 'HW_QUEUES_AT_IMPORT', None)
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '24')
    assert runtime.require_hw_queues(22) == 24
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '4')
    with pytest.raises(RuntimeError, match='GPU_MAX_HW_QUEUES'):
        runtime.require_hw_queues(22)
    assert runtime.require_hw_queues(22, allow_aliasing=True) == 4
    # HIP initialised before the import with the default queues (the ROS-node shape): the value of THAT moment counts
    monkeypatch.setattr(de6d_amd, 'HW_QUEUES_AT_IMPORT', 4)
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '24')
    with pytest.raises(RuntimeError):
        runtime.require_hw_queues(6)


def test_whole_path_scalars_put_the_driver_timed_fraction_at_the_top_of_roofline():
    """round-5 review item 6: roofline.frac is the DRIVER-TIMED whole-path fraction (algorithmic GFLOP per scene x value /
    peak); the launch-by-launch family figures and the dominant launch become top-level scalars beside it; with clock
    samples the fraction at the held clock is stated next to the nominal one"""
    from bench_legs import whole_path_scalars, MFMA_F32_PEAK_TFLOPS
    roof = {"frac": 0.659, "achieved": 103.7, "algorithmic_gflop_per_pass": 519.11, "saturated": {"frac": 0.736},
            "dominant_launch": {"frac": 0.764}}
    out = whole_path_scalars(roof, 80, 15009.82, {"sclk_mhz": 2200.0, "power_w": 1000.0})
    assert out["family_frac_idle"] == 0.659 and out["family_frac_saturated"] == 0.736 and out["dominant_launch_frac"] == 0.764
    assert abs(out["algorithmic_gflop_per_scene"] - 6.4889) < 1e-3
    assert abs(out["achieved"] - 6.488875 * 15009.82 / 1e3) < 0.01
    assert out["frac"] == out["whole_path_frac"] == round(out["achieved"] / MFMA_F32_PEAK_TFLOPS, 4) or abs(out["frac"] - 0.6192) < 2e-4
    assert out["sclk_mhz"] == 2200.0 and out["power_w"] == 1000.0
    assert abs(out["whole_path_frac_at_held_clock"] - out["frac"] * 2400.0 / 2200.0) < 1e-4
    assert all(not isinstance(out[k], dict) for k in ("whole_path_frac", "family_frac_idle", "family_frac_saturated", "dominant_launch_frac"))
