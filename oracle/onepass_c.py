"""ORACLE -- test infrastructure only.  ctypes wrapper + build recipe for onepass_c.c."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "onepass_c.c")
_LIB = os.environ.get("ONEPASS_ORACLE_LIB") or os.path.join(_HERE, "_build", "libonepass_oracle.so")
_ASAN_LIB = os.path.join(_HERE, "_build", "libonepass_oracle_asan.so")
_lib = None


def build(force=False):
    """gcc -O2 -shared -fPIC oracle/onepass_c.c -> oracle/_build/libonepass_oracle.so"""
    if (not force and os.path.exists(_LIB)
            and os.path.getmtime(_LIB) >= os.path.getmtime(_SRC)):
        return _LIB
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _LIB, _SRC])
    return _LIB


def build_sanitized(clang, force=False):
    """clang -fsanitize=address,undefined build of the same file (tests/test_sanitized_host.py loads it through
    ONEPASS_ORACLE_LIB in a child process that preloads the sanitizer runtime)."""
    if (not force and os.path.exists(_ASAN_LIB)
            and os.path.getmtime(_ASAN_LIB) >= os.path.getmtime(_SRC)):
        return _ASAN_LIB
    os.makedirs(os.path.dirname(_ASAN_LIB), exist_ok=True)
    subprocess.check_call([clang, "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                           "-shared-libsan", "-shared", "-fPIC", "-o", _ASAN_LIB, _SRC])
    return _ASAN_LIB


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        lib = ctypes.CDLL(_LIB)
        lib.oracle_extract.restype = ctypes.c_int
        lib.oracle_extract.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32] + \
                                      [ctypes.c_void_p] * 5
        lib.oracle_pairs_get.restype = ctypes.c_int
        lib.oracle_pairs_get.argtypes = [ctypes.c_void_p] * 3
        _lib = lib
    return _lib


def extract(image, max_label=None, origin=(0, 0, 0), own_first_plane=True):
    """Same result dict as oracle.onepass.extract, computed by the C restatement."""
    V = np.ascontiguousarray(image)
    if V.ndim == 2:
        V = V[:, :, None]
    if V.dtype not in (np.uint16, np.uint32):
        raise TypeError("uint16 / uint32 volumes only")
    L = int(V.max()) if max_label is None else int(max_label)
    lib = _load()
    dims = np.asarray(V.shape, dtype=np.int64)
    org = np.asarray(origin, dtype=np.int64)
    count = np.zeros(L + 1, dtype=np.uint64)
    bbox = np.zeros((L + 1, 6), dtype=np.int32)
    sum1 = np.zeros((L + 1, 3), dtype=np.uint64)
    sum2 = np.zeros((L + 1, 6), dtype=np.uint64)
    npairs = np.zeros(1, dtype=np.int64)
    rc = lib.oracle_extract(V.ctypes.data, V.dtype.itemsize, dims.ctypes.data, org.ctypes.data,
                            int(bool(own_first_plane)), L, count.ctypes.data, bbox.ctypes.data,
                            sum1.ctypes.data, sum2.ctypes.data, npairs.ctypes.data)
    if rc == -2:
        raise ValueError("label exceeds max_label %d" % L)
    if rc != 0:
        raise RuntimeError("oracle_extract failed: %d" % rc)
    n = int(npairs[0])
    lo = np.zeros(n, dtype=np.uint32)
    hi = np.zeros(n, dtype=np.uint32)
    faces = np.zeros((n, 3), dtype=np.uint64)
    lib.oracle_pairs_get(lo.ctypes.data, hi.ctypes.data, faces.ctypes.data)
    return dict(max_label=L, count=count, bbox=bbox, sum1=sum1, sum2=sum2,
                pair_lo=lo, pair_hi=hi, pair_faces=faces)


if __name__ == "__main__":
    print(build(force=True))
