"""TEST INFRASTRUCTURE — ctypes loader for the CPU oracle (oracle/liborc.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED: see oracle/*.c headers.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "liborc.so"

KP64 = np.dtype([("x", "<f8"), ("y", "<f8"), ("angle", "<f4"), ("octave", "<i4")])
KNN2 = np.dtype([("idx1", "<i4"), ("dist1", "<i4"), ("idx2", "<i4"), ("dist2", "<i4")])

_lib = None


def build(force: bool = False) -> Path:
    srcs = list(HERE.glob("*.c")) + list(HERE.glob("*.h")) + [HERE / "Makefile"]
    if force or not LIB.exists() or any(s.stat().st_mtime > LIB.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-s", "-C", str(HERE)] + (["-B"] if force else []), check=True)
    return LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        _lib = C.CDLL(str(LIB))
        _lib.orc_hamming.restype = C.c_int
        _lib.orc_bf_filter.restype = C.c_int
        _lib.orc_stereo_match.restype = C.c_int
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data if a.size else 0)


def hamming(a, b) -> int:
    a = np.ascontiguousarray(a, np.uint64)
    b = np.ascontiguousarray(b, np.uint64)
    return lib().orc_hamming(_p(a), _p(b))


def bf_knn2(q, t, threads: int = 1) -> np.ndarray:
    q = np.ascontiguousarray(q, np.uint64).reshape(-1, 4)
    t = np.ascontiguousarray(t, np.uint64).reshape(-1, 4)
    out = np.zeros(q.shape[0], KNN2)
    lib().orc_bf_knn2(_p(q), C.c_int(q.shape[0]), _p(t), C.c_int(t.shape[0]), _p(out), C.c_int(threads))
    return out


def bf_filter(knn, threshold: int, ratio: float) -> np.ndarray:
    knn = np.ascontiguousarray(knn, KNN2)
    pairs = np.zeros((max(knn.shape[0], 1), 2), np.int32)
    n = lib().orc_bf_filter(_p(knn), C.c_int(knn.shape[0]), C.c_int(threshold), C.c_float(ratio), _p(pairs))
    return pairs[:n].copy()


def stereo_match(left, dl, right, dr, bf, level_scale, relaxed=True, right_points=None, depth=None):
    left = np.ascontiguousarray(left, KP64)
    right = np.ascontiguousarray(right, KP64)
    dl = np.ascontiguousarray(dl, np.uint64).reshape(-1, 4)
    dr = np.ascontiguousarray(dr, np.uint64).reshape(-1, 4)
    nl = left.shape[0]
    rp = np.full(nl, -1000.0, np.float32) if right_points is None else np.array(right_points, np.float32)
    dp = np.full(nl, -1000.0, np.float32) if depth is None else np.array(depth, np.float32)
    ls = np.ascontiguousarray(level_scale, np.float32)
    n = lib().orc_stereo_match(_p(left), _p(dl), C.c_int(nl), _p(right), _p(dr), C.c_int(right.shape[0]),
                               C.c_double(bf), _p(ls), C.c_int(int(relaxed)), _p(rp), _p(dp))
    return n, rp, dp
