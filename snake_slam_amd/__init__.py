"""snake_slam_amd — MI355X (gfx950) implementation of Snake-SLAM's per-frame feature pipeline
(ORB extract, Hamming matchers) and local bundle adjustment, behind a C ABI
(``include/snake_hip.h``).  The Python layer only mirrors the reference's call shapes for tests
and benchmarks; all arithmetic runs in ``lib/libsnake_hip.so`` (hand-written HIP).  There is no
CPU fallback — a missing library raises ``SnakeHipError``.
"""
from ._lib import SnakeHipError, load  # noqa: F401

__all__ = ["SnakeHipError", "load"]
