"""Shared synthetic inputs for the parity tests (seeded; SURVEY.md 8d scene recipe, reduced)."""
import numpy as np


from de6d_amd.synthetic import make_scene, make_batch, beam_scene, beam_batch, points_tensor  # noqa: F401,E402


def random_boxes(seed, k, spread=30.0):
    rng = np.random.default_rng(seed)
    b = np.zeros((k, 7), np.float32)
    b[:, 0] = rng.uniform(0, spread, k)
    b[:, 1] = rng.uniform(-spread / 3, spread / 3, k)
    b[:, 2] = rng.uniform(-1, 1, k)
    b[:, 3] = rng.uniform(1, 5, k)
    b[:, 4] = rng.uniform(1, 3, k)
    b[:, 5] = rng.uniform(1, 2, k)
    b[:, 6] = rng.uniform(-4, 7, k)
    return b
