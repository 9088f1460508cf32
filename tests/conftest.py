import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The pipelined parity tests (tests/test_timed_path_gpu.py) must run in the regime bench.py times: 16 main + 6 sampler streams on
# 24 hardware queues.  ROCm reads GPU_MAX_HW_QUEUES once, when the HIP runtime initialises, so it is exported here, before any
# test module can touch the GPU (round 4's GPU run had the default 4 queues: the streams aliased and serialised).  Importing
# the package does the same for any other caller (de6d_amd/__init__.py); ScenePipeline raises when fewer queues are in effect.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
import de6d_amd  # noqa: E402,F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_ops():
    from oracle import ops
    ops.build()
    return ops
