"""include/det6d_math.h accuracy (through the oracle build) against numpy float64."""
import numpy as np


def ulp_err(got, want64):
    want32 = want64.astype(np.float32)
    sp = np.spacing(np.abs(want32)).astype(np.float64)
    return float(np.max(np.abs(got.astype(np.float64) - want64) / sp))


def test_exp_log_sigmoid(oracle_ops):
    rng = np.random.default_rng(0)
    x = rng.uniform(-87, 88, 100000).astype(np.float32)
    assert ulp_err(oracle_ops.math_fn('exp', x), np.exp(x.astype(np.float64))) <= 2.0
    x = np.exp(rng.uniform(-80, 80, 100000)).astype(np.float32)
    assert ulp_err(oracle_ops.math_fn('log', x), np.log(x.astype(np.float64))) <= 2.0
    x = rng.uniform(-15, 15, 100000).astype(np.float32)
    assert ulp_err(oracle_ops.math_fn('sigmoid', x), 1 / (1 + np.exp(-x.astype(np.float64)))) <= 3.0
    np.testing.assert_array_equal(oracle_ops.math_fn('exp', np.array([0, -200, 100, np.nan], np.float32))[:3],
                                  np.array([1, 0, np.inf], np.float32))
    assert oracle_ops.math_fn('sigmoid', np.array([0], np.float32))[0] == 0.5


def test_trig(oracle_ops):
    rng = np.random.default_rng(1)
    x = rng.uniform(-20, 20, 200000).astype(np.float32)
    assert ulp_err(oracle_ops.math_fn('sin', x), np.sin(x.astype(np.float64))) <= 2.0
    assert ulp_err(oracle_ops.math_fn('cos', x), np.cos(x.astype(np.float64))) <= 2.0
    # sin(-x) = -sin(x), cos(-x) = cos(x) exactly (check_in_box2d relies on it)
    np.testing.assert_array_equal(oracle_ops.math_fn('sin', -x), -oracle_ops.math_fn('sin', x))
    np.testing.assert_array_equal(oracle_ops.math_fn('cos', -x), oracle_ops.math_fn('cos', x))
    xx = rng.normal(size=200000).astype(np.float32)
    yy = rng.normal(size=200000).astype(np.float32)
    assert ulp_err(oracle_ops.math_fn('atan2', xx, yy), np.arctan2(yy.astype(np.float64), xx.astype(np.float64))) <= 3.5
    spec = oracle_ops.math_fn('atan2', np.array([0, -0.0, 1, -1, 0, 0], np.float32), np.array([0, 0, 0, 0, 1, -1], np.float32))
    np.testing.assert_allclose(spec, [0, np.pi, 0, np.pi, np.pi / 2, -np.pi / 2], rtol=1e-7)
