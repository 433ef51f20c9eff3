"""CPU oracle for the INSIDER hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``insider_amd``) never does.

PARITY UNPINNED: the reference ships no golden vectors for this path and cannot
be built in this image; see the header of ``insider_oracle.c``.

``c_oracle`` wraps ``_build/libinsider_oracle.so`` (C restatement in the
reference's formulation, OpenMP); ``numpy_oracle`` is an independent, slow,
readable numpy restatement used to cross-check the C one on small cases.
"""
from . import c_oracle  # noqa: F401
