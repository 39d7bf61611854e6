"""CPU oracle for the NeRF volume-rendering hot path (TEST INFRASTRUCTURE ONLY).

This package is a PyTorch-CPU fp32 restatement of the reference algorithm
(JulianKnodt/nerf_atlas, mounted read-only at /root/reference in the build
container).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product package ``nerf_atlas_amd`` never
does.  Parity is pinned: every function here is checked against fixtures under
``tests/golden/`` that were produced by importing the real reference
(``tools/gen_golden.py``).
"""
from .nerf_oracle import *  # noqa: F401,F403
