"""CPU oracle for the IS-Fusion LiDAR / fusion hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product path (``isfusion_amd``) never imports it and fails loudly without its HIP
library.  See ``oracle/isf_oracle.c`` for the per-function reference citations and pinning status.
"""
from .ref_ops import *  # noqa: F401,F403
