"""CPU oracle for the plane-sweep hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (deep3d_aerial_amd) never does, and has no CPU fallback.
"""
from .oracle import *  # noqa: F401,F403
from . import fusion  # noqa: F401,E402  (row N1: consistency check + fusion accumulators)
from .oracle import build as _build_planesweep  # noqa: E402


def build(force=False):
    """Compile both oracle libraries (idempotent)."""
    _build_planesweep(force)
    fusion.build(force)
