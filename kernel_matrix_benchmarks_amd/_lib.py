"""ctypes binding of libkmvp.so (include/kmvp.h).

The library is the product: there is NO fallback.  If it cannot be loaded, or no
GPU is present, every compute entry point raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KMVP_LIB", os.path.join(_HERE, "libkmvp.so"))

KMVP_F32, KMVP_F64, KMVP_BF16 = 0, 1, 2
STATUS = {
    0: "OK", 1: "INVALID", 2: "UNSUPPORTED", 3: "DEVICE", 4: "COMM", 5: "NOMEM", 6: "NOT_CONVERGED",
}
UNIQUE_ID_BYTES = 128

# every symbol include/kmvp.h declares: (name, restype, argtypes)
_c = ctypes
SYMBOLS = [
    ("kmvp_abi_version", _c.c_int, []),
    ("kmvp_device_count", _c.c_int, []),
    ("kmvp_create", _c.c_void_p, [_c.c_int, _c.POINTER(_c.c_int)]),
    ("kmvp_destroy", None, [_c.c_void_p]),
    ("kmvp_last_error", _c.c_char_p, [_c.c_void_p]),
    ("kmvp_set_points", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64,
                                   _c.c_int, _c.c_int, _c.c_int64, _c.c_int64]),
    ("kmvp_fit", _c.c_int, [_c.c_void_p, _c.c_int]),
    ("kmvp_set_signal", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int]),
    ("kmvp_gaussian", _c.c_int, [_c.c_void_p]),
    ("kmvp_gaussian_norm", _c.c_int, [_c.c_void_p]),
    ("kmvp_absexp", _c.c_int, [_c.c_void_p]),
    ("kmvp_absexp_norm", _c.c_int, [_c.c_void_p]),
    ("kmvp_invdist", _c.c_int, [_c.c_void_p]),
    ("kmvp_invdist_norm", _c.c_int, [_c.c_void_p]),
    ("kmvp_expdot", _c.c_int, [_c.c_void_p]),
    ("kmvp_expdot_norm", _c.c_int, [_c.c_void_p]),
    ("kmvp_get_result", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64]),
    ("kmvp_gaussian_cg_solve", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_double, _c.c_int,
                                          _c.c_void_p, _c.POINTER(_c.c_int), _c.POINTER(_c.c_double)]),
    ("kmvp_absexp_cg_solve", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_double, _c.c_int,
                                        _c.c_void_p, _c.POINTER(_c.c_int), _c.POINTER(_c.c_double)]),
    ("kmvp_invdist_minres_solve", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_double, _c.c_int,
                                             _c.c_void_p, _c.POINTER(_c.c_int), _c.POINTER(_c.c_double)]),
    ("kmvp_comm_get_unique_id", _c.c_int, [_c.c_void_p]),
    ("kmvp_comm_init", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int]),
    ("kmvp_comm_init_host", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int]),
    ("kmvp_comm_world", _c.c_int, [_c.c_void_p]),
    ("kmvp_comm_rank", _c.c_int, [_c.c_void_p]),
    ("kmvp_last_allreduce_ms", _c.c_double, [_c.c_void_p]),
    ("kmvp_set_option", _c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_int64]),
    ("kmvp_device_bytes", _c.c_int64, [_c.c_void_p]),
    ("kmvp_last_kernel_ms", _c.c_double, [_c.c_void_p]),
    ("kmvp_last_total_ms", _c.c_double, [_c.c_void_p]),
    ("kmvp_last_kernel_name", _c.c_char_p, [_c.c_void_p]),
    ("kmvp_last_dispatch_note", _c.c_char_p, [_c.c_void_p]),
]

HOST_ALLREDUCE_FN = _c.CFUNCTYPE(_c.c_int, _c.c_void_p, _c.POINTER(_c.c_double), _c.c_int64, _c.c_int)
OP_SUM, OP_MIN = 0, 1

_lib = None


class KmvpError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libkmvp: {STATUS.get(code, code)}: {message}")
        self.code = code


def load():
    """Loads libkmvp.so and types every entry point.  Raises if the library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C kernel_matrix_benchmarks_amd/csrc).  There is no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def dtype_code(precision):
    """numpy dtype / string ('float32', algos.yaml:157) / 'bfloat16' -> (kmvp_dtype, host numpy dtype).
    float16 (algos.yaml:157,160): the reference casts its inputs to float16 and lets numpy do float16 arithmetic;
    here the inputs are ROUNDED to float16 by the plugin (``input_rounding``) and the arithmetic is float32."""
    if isinstance(precision, str) and precision.lower() in ("bfloat16", "bf16"):
        return KMVP_BF16, np.dtype(np.float32)
    dt = np.dtype(precision)
    if dt == np.float16:
        return KMVP_F32, np.dtype(np.float32)
    if dt == np.float64:
        return KMVP_F64, dt
    if dt == np.float32:
        return KMVP_F32, dt
    raise NotImplementedError(f"precision {precision!r} is not supported by the MI355X backend")


def input_rounding(precision):
    """dtype the host arrays are rounded to BEFORE the cast to the working precision (None: no extra rounding)."""
    if isinstance(precision, str) and precision.lower() in ("bfloat16", "bf16"):
        return None
    return np.dtype(np.float16) if np.dtype(precision) == np.float16 else None


class Context:
    """Owns one kmvp_ctx (one GPU).  Thin, typed wrappers; all errors raise KmvpError."""

    def __init__(self, device=0):
        self._lib = load()
        status = ctypes.c_int(0)
        self._ctx = self._lib.kmvp_create(int(device), ctypes.byref(status))
        if not self._ctx:
            msg = self._lib.kmvp_last_error(None)
            raise KmvpError(status.value, msg.decode() if msg else "kmvp_create failed")
        self.device = int(device)
        self.comm_world = 0  # ranks of the communicator attached to THIS context (0: none)

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.kmvp_last_error(self._ctx)
            raise KmvpError(rc, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.kmvp_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_points(self, y, x, dtype_code_, j_offset=0, M_total=None):
        M, D = y.shape
        N = M if x is None else x.shape[0]
        if x is not None and x.shape[1] != D:
            raise ValueError(f"target points have {x.shape[1]} coordinates, source points {D}")
        self.M, self.N, self.D = M, N, D
        self._check(self._lib.kmvp_set_points(
            self._ctx, y.ctypes.data, M, None if x is None else x.ctypes.data, N, D, dtype_code_,
            int(j_offset), int(M if M_total is None else M_total)))

    def fit(self, kernel):
        self._check(self._lib.kmvp_fit(self._ctx, {"gaussian": 0, "absolute-exponential": 1, "inverse-distance": 2}[kernel]))

    def set_signal(self, b):
        if b is None:
            self._check(self._lib.kmvp_set_signal(self._ctx, None, 1))
        else:
            # the library copies M * E elements from the pointer: a shorter array must never reach it
            if b.ndim != 2 or b.shape[0] != getattr(self, "M", b.shape[0]):
                raise ValueError(f"source_signal has shape {b.shape}, expected ({getattr(self, 'M', '?')}, E): one row per "
                                 "source point")
            self._check(self._lib.kmvp_set_signal(self._ctx, b.ctypes.data, b.shape[1]))

    def run(self, kernel, normalize_rows):
        entry = {
            ("gaussian", False): self._lib.kmvp_gaussian,
            ("gaussian", True): self._lib.kmvp_gaussian_norm,
            ("absolute-exponential", False): self._lib.kmvp_absexp,
            ("absolute-exponential", True): self._lib.kmvp_absexp_norm,
            ("inverse-distance", False): self._lib.kmvp_invdist,
            ("inverse-distance", True): self._lib.kmvp_invdist_norm,
            ("exp-dot", False): self._lib.kmvp_expdot,
            ("exp-dot", True): self._lib.kmvp_expdot_norm,
        }[(kernel, bool(normalize_rows))]
        self._check(entry(self._ctx))

    def get_result(self, N, E):
        out = np.empty((N, E), dtype=np.float64)
        self._check(self._lib.kmvp_get_result(self._ctx, out.ctypes.data, out.size))
        return out

    def cg_solve(self, kernel, a, rtol, maxit):
        if a.ndim != 2 or a.shape[0] != getattr(self, "N", a.shape[0]):
            # the Krylov vectors have one row per point (all points are targets); the library reads N * E elements
            raise ValueError(f"target_signal has shape {a.shape}, the point cloud has {self.N} points")
        entry = {
            "gaussian": self._lib.kmvp_gaussian_cg_solve,
            "absolute-exponential": self._lib.kmvp_absexp_cg_solve,
            "inverse-distance": self._lib.kmvp_invdist_minres_solve,  # indefinite: MINRES
        }.get(kernel)
        if entry is None:
            raise NotImplementedError(f"no solver for kernel {kernel}")
        M, E = a.shape
        out = np.empty((M, E), dtype=np.float64)
        iters = ctypes.c_int(0)
        resid = ctypes.c_double(0.0)
        rc = entry(self._ctx, a.ctypes.data, E, float(rtol), int(maxit), out.ctypes.data,
                   ctypes.byref(iters), ctypes.byref(resid))
        if rc not in (0, 6):
            self._check(rc)
        return out, iters.value, resid.value, rc == 0

    def comm_init(self, unique_id, rank, world):
        buf = (ctypes.c_char * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._lib.kmvp_comm_init(self._ctx, buf, int(rank), int(world)))
        self.comm_world = int(world)  # attachment is recorded on the context itself (sharding.Communicator.attach)

    def comm_init_host(self, allreduce, rank, world):
        """REHEARSAL transport (include/kmvp.h kmvp_comm_init_host): ``allreduce(array, op)`` must reduce a float64
        numpy array in place over all ranks (op: OP_SUM or OP_MIN).  The library stages the exchange buffer through
        host memory and calls it."""

        def trampoline(_user, buf, count, op):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(int(count),)), int(op))
                return 0
            except Exception as e:  # never let a Python exception cross the C frame
                self._host_error = e
                return 1

        self._host_cb = HOST_ALLREDUCE_FN(trampoline)  # kept alive as long as the context
        self._check(self._lib.kmvp_comm_init_host(self._ctx, ctypes.cast(self._host_cb, ctypes.c_void_p), None,
                                                  int(rank), int(world)))
        self.comm_world = int(world)

    @property
    def rccl_ranks(self):
        """Ranks the attached communicator itself reports (1 without one)."""
        return int(self._lib.kmvp_comm_world(self._ctx))

    @property
    def last_allreduce_ms(self):
        return float(self._lib.kmvp_last_allreduce_ms(self._ctx))

    def set_option(self, key, value):
        self._check(self._lib.kmvp_set_option(self._ctx, key.encode(), int(value)))

    @property
    def device_bytes(self):
        return int(self._lib.kmvp_device_bytes(self._ctx))

    @property
    def last_kernel_ms(self):
        return float(self._lib.kmvp_last_kernel_ms(self._ctx))

    @property
    def last_total_ms(self):
        return float(self._lib.kmvp_last_total_ms(self._ctx))

    @property
    def last_kernel_name(self):
        return self._lib.kmvp_last_kernel_name(self._ctx).decode()

    @property
    def last_dispatch_note(self):
        return self._lib.kmvp_last_dispatch_note(self._ctx).decode()


def comm_unique_id():
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    rc = load().kmvp_comm_get_unique_id(buf)
    if rc != 0:
        msg = load().kmvp_last_error(None)
        raise KmvpError(rc, msg.decode() if msg else "")
    return bytes(buf.raw)


def device_count():
    n = load().kmvp_device_count()
    return max(0, n)
