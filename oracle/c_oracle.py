"""ctypes wrapper of ``oracle/kmvp_oracle.c`` -- TEST INFRASTRUCTURE ONLY (see the
header of that file).  Used by tests, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by the product path."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkmvp_oracle.so")
_KERNEL_ID = {"gaussian": 0, "absolute-exponential": 1, "inverse-distance": 2}
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "kmvp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libkmvp_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        for name in ("kmvp_oracle_product_f64", "kmvp_oracle_product_f32"):
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [
                ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
            ]
        _lib.kmvp_oracle_threads.restype = ctypes.c_int
    return _lib


def threads():
    return int(lib().kmvp_oracle_threads())


def product(
    *, kernel, source_points, target_points=None, source_signal=None, normalize_rows=False,
    density_estimation=False, precision=np.float64, rows=None, j_offset=0, M_total=None,
    raw_sums=False,
):
    """Same contract as ``kmvp_oracle.product`` (slow squared-distance form only)."""
    precision = np.dtype(precision)
    y = np.ascontiguousarray(source_points, dtype=precision)
    x = y if target_points is None else np.ascontiguousarray(target_points, dtype=precision)
    M, D = y.shape
    N = x.shape[0]
    b = None
    E = 1
    if not density_estimation and source_signal is not None:
        b = np.ascontiguousarray(source_signal, dtype=precision)
        E = b.shape[1]
    if rows is not None:
        rows = np.ascontiguousarray(rows, dtype=np.int64)
    n = N if rows is None else rows.shape[0]
    num = np.empty((n, E), dtype=np.float64)
    den = np.empty((n, 1), dtype=np.float64)
    fn = lib().kmvp_oracle_product_f64 if precision == np.float64 else lib().kmvp_oracle_product_f32
    rc = fn(
        _KERNEL_ID[kernel], y.ctypes.data, M, x.ctypes.data, N, D,
        None if b is None else b.ctypes.data, E, None if rows is None else rows.ctypes.data, n,
        j_offset, M if M_total is None else M_total, num.ctypes.data, den.ctypes.data,
    )
    if rc != 0:
        raise RuntimeError(f"kmvp_oracle_product failed ({rc})")
    if raw_sums:
        return num, den
    if normalize_rows:
        if b is None:
            return np.ones((n, 1))
        return num / den
    return num
