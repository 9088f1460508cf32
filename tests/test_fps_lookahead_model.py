"""The multi-pick samplers' decision rule is sound under any schedule (model: tests/models/fps_lookahead.py; kernels:
csrc/fps_coop.hip, csrc/fps_seq.hip)."""
import numpy as np
import pytest

from tests.models import fps_lookahead as M


def _regions(n, nreg, rng, spatial, xyz):
    if spatial:   # compact regions along x, like the Morton regions of the kernel
        order = np.argsort(xyz[:, 0], kind='stable')
    else:
        order = rng.permutation(n)
    return np.array_split(order, nreg)


@pytest.mark.parametrize("case", ["uniform", "lattice", "duplicates", "all_equal", "clustered"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_lookahead_model_equals_sequential_fps(case, seed):
    rng = np.random.default_rng(100 + seed)
    n, m, nreg = 512, 96, 8
    if case == "uniform":
        xyz = rng.uniform(-10, 10, (n, 3))
    elif case == "lattice":          # exact ties everywhere
        xyz = rng.integers(0, 6, (n, 3)).astype(np.float64)
    elif case == "duplicates":       # the reference pads short frames with repeated points
        base = rng.uniform(-10, 10, (n // 4, 3))
        xyz = base[rng.integers(0, n // 4, n)]
    elif case == "all_equal":
        xyz = np.ones((n, 3)) * 3.25
    else:
        xyz = np.concatenate([rng.normal(c, 0.3, (n // 4, 3)) for c in ((0, 0, 0), (8, 0, 0), (0, 8, 0), (30, 30, 0))])
    xyz = xyz.astype(np.float32)
    want = M.fps_sequential(xyz, m)
    for spatial in (True, False):
        regs = _regions(n, nreg, rng, spatial, xyz)
        stats = {}
        got = M.run(xyz, m, regs, seed=seed * 7 + spatial, stats=stats)
        assert got == want, (case, spatial)


def test_lookahead_model_two_point_regions_and_single_region():
    rng = np.random.default_rng(5)
    xyz = rng.integers(0, 4, (64, 3)).astype(np.float32)
    want = M.fps_sequential(xyz, 64)
    assert M.run(xyz, 64, np.array_split(np.arange(64), 32), seed=3) == want     # regions of exactly two points
    assert M.run(xyz, 64, [np.arange(64)], seed=4) == want


@pytest.mark.parametrize("depth", [1, 2, 4])
@pytest.mark.parametrize("case", ["uniform", "lattice", "duplicates"])
def test_lockstep_and_delayed_schedules(case, depth):
    """the kernels' own schedule (lockstep rounds: `greedy`) and the delayed-record schedule, at the candidate depths the
    kernels use (4; 2 and 1 for the record), give plain FPS; a round of the lockstep schedule makes at least one pick"""
    rng = np.random.default_rng(7)
    n, m, nreg = 512, 128, 16
    if case == "uniform":
        xyz = rng.uniform(-10, 10, (n, 3))
    elif case == "lattice":
        xyz = rng.integers(0, 6, (n, 3)).astype(np.float64)
    else:
        base = rng.uniform(-10, 10, (n // 4, 3))
        xyz = base[rng.integers(0, n // 4, n)]
    xyz = xyz.astype(np.float32)
    want = M.fps_sequential(xyz, m)
    regs = _regions(n, nreg, rng, True, xyz)
    st = {}
    assert M.run(xyz, m, regs, greedy=True, depth=depth, stats=st) == want
    assert st['blocks'] <= m - 1                       # every round decided something
    if depth > 1 and case == "uniform":
        assert st['blocks'] < (m - 1) // 2             # ... and on ordinary clouds several picks
    assert M.run(xyz, m, regs, delay=3, depth=depth) == want


@pytest.mark.parametrize("case", ["duplicates", "lattice", "all_equal", "every_point_twice"])
def test_hidden_duplicates_change_nothing(case):
    """sq_hide_lane_duplicates: points that repeat an order-earlier point of their region start at min-distance 0 (out of the
    records); the picks stay plain FPS under the lockstep, the delayed and a random schedule — including m > the number of
    distinct points, where the reference re-picks"""
    rng = np.random.default_rng(11)
    n, m, nreg = 512, 160, 16
    if case == "duplicates":
        base = rng.uniform(-10, 10, (n // 4, 3))
        xyz = base[rng.integers(0, n // 4, n)]
    elif case == "lattice":
        xyz = rng.integers(0, 4, (n, 3)).astype(np.float64)          # 64 distinct points: m exceeds them
    elif case == "all_equal":
        xyz = np.ones((n, 3)) * 3.25
    else:
        half = rng.uniform(-10, 10, (n // 2, 3))
        xyz = np.concatenate([half, half])
    xyz = xyz.astype(np.float32)
    want = M.fps_sequential(xyz, m)
    for spatial in (True, False):
        regs = _regions(n, nreg, rng, spatial, xyz)
        assert M.run(xyz, m, regs, greedy=True, depth=4, hide_duplicates=True, stats={}) == want
        assert M.run(xyz, m, regs, delay=2, depth=2, hide_duplicates=True, seed=3) == want
        assert M.run(xyz, m, regs, seed=5, hide_duplicates=True) == want


@pytest.mark.parametrize("case", ["uniform", "duplicates", "lattice", "all_equal", "every_point_twice"])
@pytest.mark.parametrize("depth", [2, 4])
def test_kernel_key_scheme(case, depth):
    """the kernels' way of taking the decision — per candidate a threshold key and a fallback key, ONE maximum over all
    candidates, no reduction per region (fps_seq.hip) — under the kernels' lockstep schedule and the delayed one, ties and
    hidden duplicates included; a lockstep round always makes a pick"""
    rng = np.random.default_rng(23)
    n, m, nreg = 512, 200, 16
    if case == "uniform":
        xyz = rng.uniform(-10, 10, (n, 3))
    elif case == "duplicates":
        base = rng.uniform(-10, 10, (n // 4, 3))
        xyz = base[rng.integers(0, n // 4, n)]
    elif case == "lattice":
        xyz = rng.integers(0, 4, (n, 3)).astype(np.float64)
    elif case == "all_equal":
        xyz = np.ones((n, 3)) * 3.25
    else:
        half = rng.uniform(-10, 10, (n // 2, 3))
        xyz = np.concatenate([half, half])
    xyz = xyz.astype(np.float32)
    want = M.fps_sequential(xyz, m)
    for spatial in (True, False):
        regs = _regions(n, nreg, rng, spatial, xyz)
        for hide in (False, True):
            st = {}
            assert M.run(xyz, m, regs, greedy=True, depth=depth, kernel_keys=True, hide_duplicates=hide, stats=st) == want
            assert st['blocks'] <= m - 1
            assert M.run(xyz, m, regs, delay=2, depth=depth, kernel_keys=True, hide_duplicates=hide, seed=9) == want


@pytest.mark.parametrize("case", ["uniform", "lattice", "every_point_twice"])
def test_lists_of_varying_length(case):
    """records may list 1 .. 4 candidates, each record a length of its own (the kernels ask for short lists after single-pick
    rounds): still plain FPS, with the kernels' key scheme and hidden duplicates"""
    rng = np.random.default_rng(31)
    n, m, nreg = 512, 180, 16
    if case == "uniform":
        xyz = rng.uniform(-10, 10, (n, 3))
    elif case == "lattice":
        xyz = rng.integers(0, 4, (n, 3)).astype(np.float64)
    else:
        half = rng.uniform(-10, 10, (n // 2, 3))
        xyz = np.concatenate([half, half])
    xyz = xyz.astype(np.float32)
    want = M.fps_sequential(xyz, m)
    regs = _regions(n, nreg, rng, True, xyz)
    for seed in (1, 2):
        assert M.run(xyz, m, regs, greedy=True, depth=0, kernel_keys=True, hide_duplicates=True, stats={}, seed=seed) == want
        assert M.run(xyz, m, regs, delay=2, depth=0, kernel_keys=True, seed=seed) == want
        assert M.run(xyz, m, regs, depth=0, seed=seed) == want


@pytest.mark.parametrize("case", ["uniform", "lattice_equal_weights", "duplicates", "few_weight_levels"])
@pytest.mark.parametrize("kernel_keys", [False, True])
def test_score_weighted_rule_equals_sequential_weighted_fps(case, kernel_keys):
    """round 6, csrc/fps_seq.hip: fps_seq_w_kernel — the same rounds on SCORES (min-distance x weight): records ranked by score
    with min-distance and weight beside it, the sequencer's candidates tracked by min-distance, the unknown bound = the last
    candidate's old score, the owners' skip test on the region's maximal MIN-DISTANCE, pick 0 = arg-max of the weights under the
    reference's order.  Random, lockstep and delayed schedules, hidden duplicates, the kernels' key scheme."""
    rng = np.random.default_rng(11)
    n, m, nreg = 512, 128, 16
    if case == "uniform":
        xyz = rng.uniform(-10, 10, (n, 3))
        w = 1.0 / (1.0 + np.exp(-rng.normal(size=n) * 2))
    elif case == "lattice_equal_weights":          # exact score ties everywhere, pick 0 decided by the order alone
        xyz = rng.integers(0, 6, (n, 3)).astype(np.float64)
        w = np.full(n, 0.75)
    elif case == "duplicates":                     # repeated points, some with the same weight, some with another
        base = rng.uniform(-10, 10, (n // 4, 3))
        src = rng.integers(0, n // 4, n)
        xyz = base[src]
        w = np.where(rng.random(n) < 0.5, 0.5, 1.0 / (1.0 + np.exp(-rng.normal(size=n))))
    else:                                          # a handful of weight levels: ties between points of different regions
        xyz = rng.uniform(-10, 10, (n, 3))
        w = rng.choice([0.125, 0.25, 0.5, 1.0], n)
    xyz, w = xyz.astype(np.float32), w.astype(np.float32)
    want = M.fps_sequential(xyz, m, weights=w)
    assert want[0] == M.first_pick(n, M.opt_log2s(n), w)
    regs = _regions(n, nreg, rng, True, xyz)
    for kw in (dict(seed=3), dict(seed=4, greedy=True, depth=4), dict(seed=5, greedy=True, depth=0, hide_duplicates=True),
               dict(seed=6, delay=3, depth=2)):
        assert M.run(xyz, m, regs, weights=w, kernel_keys=kernel_keys, **kw) == want, (case, kw)
