"""what-if (NOT a result path, results stay correct): bench.py's worker with selected GEMM-family launches issued TWICE (they
are idempotent: plain stores or integer max into an output that already holds the value) -> how much of a kernel's stand-alone
duration the pipelined pass period really pays for.  If the period grows by the kernel's stand-alone time, the pipeline is bound
by the SUM of the family's durations and making that kernel faster pays in full; if it grows by much less, the kernel's time is
hidden behind the others.
usage: whatif_twice.py none|rows|chain|small|linear|dominant [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
mode = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
import bench
from de6d_amd import _lib as L
_real = L.call


def twice(name, *args):
    rep = False
    if name == 'det6d_mlp_rows':
        rep = mode in ('rows', 'small')
    elif name == 'det6d_mlp_chain3_compact':
        rep = mode in ('chain', 'small')
    elif name == 'det6d_mlp_group3':
        c1, c3 = args[7], args[15]
        rep = (mode == 'small' and c3 == 256 and c1 == 128 and args[11] == 128) or (mode == 'dominant' and c3 == 1024)
    elif name == 'det6d_linear':
        rep = mode == 'linear'
    if rep:
        _real(name, *args)
    return _real(name, *args)


if mode != 'none':
    L.call = twice
bench.main()
