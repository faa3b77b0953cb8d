"""ctypes loader for oracle/c (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_c.so")
_lib = None


def build(force: bool = False):
    """Compile oracle/c/lfilter.c with gcc (seconds)."""
    src = os.path.join(_HERE, "c", "lfilter.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c")])
    return _SO


def available() -> bool:
    return _load() is not None


def _load():
    global _lib
    if _lib is None and os.path.exists(_SO):
        lib = ctypes.CDLL(_SO)
        i64, p = ctypes.c_int64, ctypes.c_void_p
        lib.lfilter_df1_f32.argtypes = [p, i64, i64, p, p, p]
        lib.lfilter_df1_f32.restype = None
        lib.block_energy_f32.argtypes = [p, i64, i64, i64, i64, i64, p]
        lib.block_energy_f32.restype = None
        _lib = lib
    return _lib


def lfilter_df1_f32(x: np.ndarray, b, a) -> np.ndarray:
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, T = x.shape
    b = np.ascontiguousarray(b, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    y = np.empty_like(x)
    lib.lfilter_df1_f32(x.ctypes.data, rows, T, b.ctypes.data, a.ctypes.data, y.ctypes.data)
    return y


def block_energy_f32(y: np.ndarray, K: int, S: int, nblk: int) -> np.ndarray:
    lib = _load()
    y = np.ascontiguousarray(y, dtype=np.float32)
    rows, T = y.shape
    z = np.empty((rows, nblk), dtype=np.float32)
    lib.block_energy_f32(y.ctypes.data, rows, T, K, S, nblk, z.ctypes.data)
    return z
