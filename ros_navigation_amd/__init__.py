"""ros_navigation_amd -- MI355X-native occupancy-grid planning engine (HIMM + VFH+ + A*/RRT hot path of
jmloveyj/ros_navigation's move_control).  The product is librna.so (hand-written HIP kernels behind
the C ABI of include/rna.h); this package only holds the ctypes plumbing and the synthetic
workload generators shared by tests/ and bench.py.  See DESIGN.md."""
from . import capi, synth  # noqa: F401
from .capi import Engine, RnaError  # noqa: F401

__all__ = ["capi", "synth", "Engine", "RnaError"]
