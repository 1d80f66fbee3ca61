"""CPU oracle for the pose-refinement hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker / timed baseline -- never as the product path.
See oracle/reference_port.py for the parity-pinning statement.
"""
from .reference_port import *  # noqa: F401,F403
from . import silhouette_port  # noqa: F401,E402
