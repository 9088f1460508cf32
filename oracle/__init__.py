"""CPU oracle for the Det6D hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Nothing under ``de6d_amd/`` does (tests/test_boundary.py enforces it).

``oracle.ops`` wraps ``libdet6d_oracle.so`` (built from ``det6d_oracle.c`` by ``oracle/Makefile``)
with numpy in / numpy out functions whose names and argument order follow ``include/det6d_ops.h``.
``oracle.ref`` wraps ``_ref/libref_iou3d_cpu.so``, the reference's own ``iou3d_cpu.cpp`` compiled
from ``/root/reference`` where it lies (present only when built in the authoring container).
"""
from . import ops  # noqa: F401
