"""CPU oracle (TEST INFRASTRUCTURE ONLY): ctypes view of oracle/libfcpp_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from .oracle import *  # noqa: F401,F403
