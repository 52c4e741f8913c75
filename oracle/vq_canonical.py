"""ctypes front-end of oracle/vq_canonical.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvq_canonical.so")
_lib = None


def build():
    r = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.vq_canonical_argmin.restype = None
        _lib.vq_canonical_argmin.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def argmin(z: np.ndarray, E: np.ndarray):
    """-> (idx int64 [M], dmin float32 [M]) in the canonical evaluation order."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    E = np.ascontiguousarray(E, dtype=np.float32)
    M, D = z.shape
    K = E.shape[0]
    assert E.shape[1] == D
    idx = np.empty(M, dtype=np.int64)
    dmin = np.empty(M, dtype=np.float32)
    ee = np.empty(K, dtype=np.float32)
    _load().vq_canonical_argmin(z.ctypes.data, D, E.ctypes.data, M, K, D, idx.ctypes.data, dmin.ctypes.data, ee.ctypes.data)
    return idx, dmin
