"""CPU oracle for the plane-sweep hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (deep3d_aerial_amd) never does, and has no CPU fallback.
"""
from .oracle import *  # noqa: F401,F403
