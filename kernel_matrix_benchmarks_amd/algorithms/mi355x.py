"""MI355X plugins: the on-the-fly kernel product and the CG solver built on it.

Drop-in counterparts of the reference's ``BruteForceProductBLAS``
(bruteforce.py:61-153) and ``BruteForceSolverLAPACK`` (bruteforce.py:156-207):
same constructor keywords, same method order, same result type.  All arithmetic
happens in ``libkmvp.so`` (hand-written HIP kernels for gfx950) behind the C ABI
of ``include/kmvp.h``.  There is no CPU path: without the library or a GPU the
calls raise.

Differences a user should know about
 * ``fit()`` never forms the N x M matrix; it builds what the points alone determine
   (``kmvp_fit``: grid, cell order and tile lists of the Gaussian cell kernels, a few ms
   at 1e6 points; nothing for the other kernels).  Compare on build_time + query_time,
   which is the harness' default axis (plot.py:128).
 * The HIP context is created in ``prepare_data`` -- never at import or
   construction -- because the harness imports plugins in its parent process
   before forking the worker (main.py:262-308).
 * The solver is conjugate gradients on the product operator with a residual
   stopping rule; the dense ``lstsq`` of the reference cannot be matched
   vector-for-vector on these numerically singular matrices, so it
   reports (and is tested on) the relative residual.
"""
import numpy as np

from kernel_matrix_benchmarks_amd import _lib
from kernel_matrix_benchmarks_amd.algorithms.base import BaseProduct, BaseSolver
from kernel_matrix_benchmarks_amd import sharding

SUPPORTED_KERNELS = ("gaussian", "absolute-exponential", "inverse-distance", "exp-dot")
# "exp-dot": k(x, y) = exp(<x, y>), the transformer attention kernel the reference's README defines
# (README.md:51-59) but its plugins do not implement (bruteforce.py:18-22 has the other three): parity
# unpinned, checked against a direct numpy evaluation only (the tests' exp_dot_product).  Two routes:
#  * NATIVE (float32 / float16 inputs at D <= 64, bfloat16 at D <= 141; include/kmvp.h kmvp_expdot[_norm]): S = X Y^T
#    straight from the matrix cores, kernel values relative to a per-target running exponent (the flash-attention
#    recurrence), partial sums merged as (mantissa, exponent) pairs -- softmax attention has NO range limit on
#    <x, y>; plain products overflow where exp(<x, y>) leaves float64, as numpy's would.
#  * IDENTITY (float64, float32 at D > 64, and the solver):
#        exp(<x, y>) = exp(|x|^2 / 2) * exp(-|x/sqrt2 - y/sqrt2|^2) * exp(|y|^2 / 2)
#    i.e. a GAUSSIAN product on the points / sqrt(2) with the signal weighted by w_j = exp(|y_j|^2/2 - c)
#    (c = max_j |y_j|^2/2, so w <= 1) -- every Gaussian kernel of the library serves it.  Row-normalised: (K (w b)) /
#    (K w), the factor of the target cancels; plain products multiply it back in float64.  Valid while
#    |y_j|^2/2 spans less than the exponent range of the working precision: checked in prepare_data
#    (EXPDOT_IDENTITY_SPREAD), NotImplementedError beyond it -- never a silent zero weight.
SQRT_HALF = 0.7071067811865476
EXPDOT_NATIVE_MAX_D = 64
EXPDOT_NATIVE_MAX_D_BF16 = 141  # 16 * 9 k-steps - 3 operand columns (kmvp_mfma.hpp MFMA_DOT_AUG)
# largest spread max_j |y_j|^2/2 - min_j |y_j|^2/2 the identity route accepts: the smallest weight is exp(-spread);
# float32 / bfloat16 (8-bit exponent): e^-80 = 2^-115 is still a normal number; float64: e^-700
EXPDOT_IDENTITY_SPREAD = {"float32": 80.0, "float64": 700.0}


def _sq_norms_on_device(p_scaled, bf16):
    """|p'|^2 in float64 of the SCALED points exactly as the device will see them (float32 / float64 values as cast;
    bfloat16: rounded to nearest even as the packing kernels round) -- so that the weights exp(|y'|^2) and the
    Gaussian exp(-|x' - y'|^2) are about the same points and the identity holds to rounding of the arithmetic only."""
    q = np.asarray(p_scaled)
    if bf16:
        u = np.ascontiguousarray(q, dtype=np.float32).view(np.uint32)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
        q = u.view(np.float32)
    q = q.astype(np.float64)
    return np.sum(q * q, axis=1)


def _precision_name(precision):
    if isinstance(precision, str):
        return precision
    return np.dtype(precision).name


class MI355XProduct(BaseProduct):
    """a_i = sum_j k(x_i, y_j) b_j on one MI355X, or on several with the sources
    sharded over ranks (``comm`` = a ``sharding.Communicator``)."""

    def __init__(self, *, kernel, dimension, normalize_rows=False, precision=np.float32,
                 fast_sqdists=None, device=0, comm=None, targets_per_lane=0, feed=None, segments=0,
                 chunk=0, fast_tiles=0):
        super().__init__(kernel=kernel, dimension=dimension, normalize_rows=normalize_rows,
                         precision=precision)
        if kernel not in SUPPORTED_KERNELS:
            # same failure mode as bruteforce.py:82-85
            raise NotImplementedError(f"MI355XProduct doesn't support kernel {kernel}.")
        self._dot = kernel == "exp-dot"
        self._dot_native = False  # decided in prepare_data (needs D)
        self._device_kernel_fn = "gaussian" if self._dot else kernel
        self._dtype_code, self._host_dtype = _lib.dtype_code(precision)  # NotImplementedError if unknown
        # precision="float16" (algos.yaml:157): inputs rounded to float16 as the reference casts them
        # (bruteforce.py:103-111,126), arithmetic in float32 on the GPU
        self._round = _lib.input_rounding(precision)
        self.device = device
        self.comm = comm
        # fast_sqdists mirrors the reference's flag (bruteforce.py:70,36-49): True = expanded
        # |x|^2+|y|^2-2x.y form around one centre (matrix cores), False = difference form,
        # "centred" = expanded around per-group centres with exact recomputation of the closest
        # pairs; "cells" = the Gaussian's exp() range-reduced by grid cells, polynomial remainder
        # on the matrix cores (D <= 3: cellmm_kernel, which also sums over the sources in the MFMA
        # accumulator, where it applies; "cells-valu" = always cell_kernel, which sums on the VALU);
        # None lets the library pick the cheapest form that is as accurate as the difference form.
        if fast_sqdists not in (None, False, True, "centred", "cells", "cells-valu"):
            raise ValueError("fast_sqdists must be None, False, True, 'centred', 'cells' or 'cells-valu'")
        self.fast_sqdists = fast_sqdists
        self._options = dict(targets_per_lane=targets_per_lane, feed=feed, segments=segments,
                             chunk=chunk, fast_tiles=fast_tiles)
        self._ctx = None
        self.res = None
        self.name = f"MI355XProduct({_precision_name(precision)})" if fast_sqdists is None else (
            f"MI355XProduct({_precision_name(precision)}, fast_sqdists={fast_sqdists})")

    def _cast(self, a):
        if self._round is not None:
            a = np.asarray(a, dtype=self._round)
        return np.ascontiguousarray(a, dtype=self._host_dtype)

    # -- untimed -------------------------------------------------------------------
    def prepare_data(self, *, source_points, target_points, same_points=False,
                     density_estimation=False):
        # h5py hands numpy.bool_ attributes over (runner.py:41-43)
        self.same_points = bool(same_points)
        self.density_estimation = bool(density_estimation)
        self._dot_note = ""
        if self._dot:
            d_in = np.asarray(source_points).shape[1]
            self._dot_native = ((self._dtype_code == _lib.KMVP_F32 and d_in <= EXPDOT_NATIVE_MAX_D) or
                                (self._dtype_code == _lib.KMVP_BF16 and d_in <= EXPDOT_NATIVE_MAX_D_BF16))
            self._device_kernel_fn = "exp-dot" if self._dot_native else "gaussian"
        if self._dot and not self._dot_native:
            # the caller's points in the working precision first (what the kernel is defined on), then the identity
            ys = np.asarray(self._cast(source_points), dtype=np.float64)
            xs = ys if self.same_points else np.asarray(self._cast(target_points), dtype=np.float64)
            source_points = ys * SQRT_HALF
            target_points = xs * SQRT_HALF
            bf16 = self._dtype_code == _lib.KMVP_BF16
            hy = _sq_norms_on_device(self._cast(source_points), bf16)   # |y/sqrt2|^2 = |y|^2 / 2
            hx = hy if self.same_points else _sq_norms_on_device(self._cast(target_points), bf16)
            shift = float(np.max(hy)) if len(hy) else 0.0
            spread = float(shift - np.min(hy)) if len(hy) else 0.0
            budget = EXPDOT_IDENTITY_SPREAD["float64" if self._dtype_code == _lib.KMVP_F64 else "float32"]
            self._dot_note = (f"exp-dot through the Gaussian identity ({_precision_name(self.precision)}, D = {ys.shape[1]}): "
                              f"|y|^2/2 spans {spread:.3g} of the {budget:g} this precision's exponent range allows")
            if not spread <= budget:
                raise NotImplementedError(
                    f"exp-dot through the Gaussian identity: |y_j|^2/2 spans {spread:.4g} > {budget:g}; sources of small norm "
                    f"would silently get weight 0 in {_precision_name(self.precision)}.  Use precision='float32' with D <= "
                    f"{EXPDOT_NATIVE_MAX_D} (native online-max kernel, no range limit) or float64 (spread <= "
                    f"{EXPDOT_IDENTITY_SPREAD['float64']:g}).")
            self._w = np.exp(hy - shift).reshape(-1, 1)   # source weights, <= 1
            self._hx = hx + shift                         # log of the target factor exp(|x|^2/2 + c)
        y = self._cast(source_points)
        self.M, self.D = y.shape
        if self.same_points:
            x = None
            self.N = self.M
        else:
            x = self._cast(target_points)
            self.N = x.shape[0]
        if self._ctx is None:
            self._ctx = _lib.Context(self.device)
            for key, value in self._options.items():
                if value:
                    self._ctx.set_option(key, value)
            if self.fast_sqdists is not None:
                code = {"centred": 2, "cells": 3, "cells-valu": 4}.get(self.fast_sqdists, int(bool(self.fast_sqdists)))
                self._ctx.set_option("fast_sqdists", code)
        world = 1 if self.comm is None else self.comm.world
        if world > 1:
            # every rank keeps all targets and one contiguous slice of the sources
            self._shard = sharding.shard_range(self.M, self.comm.rank, world)
            lo, hi = self._shard
            # Gaussian (no index-based rule): shard the sources cell by cell instead of in the caller's
            # order -- every rank derives the same permutation from the full cloud it was handed
            self._order = (sharding.spatial_order(y) if self._device_kernel_fn == "gaussian" and self._host_dtype == np.float32
                           else None)
            if self._order is not None:
                targets = y if x is None else x
                y = y[self._order]
                x = targets
            self.comm.attach(self._ctx)
            # the context sees the targets as an explicit array; tell it when they ARE the
            # (unsharded) sources, which the inverse-distance zero rule of the centred form needs
            self._ctx.set_option("same_points_global", 1 if x is None else 0)
            self._ctx.set_points(np.ascontiguousarray(y[lo:hi]), y if x is None else x,
                                 self._dtype_code, j_offset=lo, M_total=self.M)
        else:
            self._shard = (0, self.M)
            self._order = None
            self._ctx.set_points(y, x, self._dtype_code)

    def prepare_query(self, *, source_signal):
        if self._dot_native and self.density_estimation:
            source_signal = np.ones((self.M, 1))  # the library's exp(<x,y>) entry takes an explicit signal
        if self._dot and not self._dot_native:
            # weighted signal [w b | w]: numerator columns and (normalised rows) the denominator column
            b = np.ones((self.M, 1)) if self.density_estimation else np.asarray(self._cast(source_signal), dtype=np.float64)
            if b.ndim == 1:
                b = b.reshape(-1, 1)
            if b.ndim != 2 or b.shape[0] != self.M:
                raise ValueError(f"source_signal has shape {b.shape}, expected ({self.M}, E)")
            self.E = b.shape[1]
            cols = [self._w * b] + ([self._w] if self.normalize_rows else [])
            wb = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype=self._host_dtype)
            lo, hi = self._shard
            if self._order is not None:
                wb = wb[self._order]
            self._ctx.set_signal(np.ascontiguousarray(wb[lo:hi]))
            return
        if self.density_estimation and not self._dot_native:
            self._ctx.set_signal(None)
            self.E = 1
            return
        b = self._cast(source_signal)
        if b.ndim == 1:
            b = b.reshape(-1, 1)
        if b.ndim != 2 or b.shape[0] != self.M:
            # the reference fails in its matmul (bruteforce.py:150); here a short array would be read past its end
            raise ValueError(f"source_signal has shape {b.shape}, expected ({self.M}, E)")
        self.E = b.shape[1]
        lo, hi = self._shard
        if self._order is not None:
            b = b[self._order]  # the sources were sharded in cell order (prepare_data)
        self._ctx.set_signal(np.ascontiguousarray(b[lo:hi]))

    def get_result(self):
        if self._dot and not self._dot_native:
            if self.normalize_rows:
                res = self._ctx.get_result(self.N, self.E + 1)
                return np.ascontiguousarray(res[:, : self.E] / res[:, self.E:])  # exp(|x_i|^2/2) cancels
            res = self._ctx.get_result(self.N, self.E)
            hx = self._hx.reshape(-1, 1)
            with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
                out = res * np.exp(hx)
                # exp(|x|^2/2 + c) alone may leave float64 where the product does not (the Gaussian factor is tiny
                # there): those rows are put together in the log domain.  (A Gaussian factor that underflowed to 0 under
                # an infinite target factor stays NaN: nothing is known about that entry, and 0 would be a guess.)
                late = ~np.isfinite(out) & np.isfinite(res) & (res != 0)
                if late.any():
                    out = np.where(late, np.sign(res) * np.exp(np.log(np.abs(res)) + hx), out)
            return np.ascontiguousarray(out)
        return self._ctx.get_result(self.N, self.E)

    # -- timed ---------------------------------------------------------------------
    def fit(self):
        """The kernel matrix is never formed; what the points alone determine (grid, cell order and tile
        lists of the Gaussian cell kernels) is built here, as the reference builds its structure in fit()."""
        if not self._dot_native:  # (the native exp(<x,y>) path has nothing the points alone determine)
            self._ctx.fit(self._device_kernel_fn)

    def query(self):
        # synchronous: the device (and the all-reduce) is done when this returns
        self._ctx.run(self._device_kernel_fn, self.normalize_rows and (self._dot_native or not self._dot))
        self.res = None  # the result stays on the device until get_result()

    # -- bookkeeping ---------------------------------------------------------------
    def set_query_arguments(self, **kwargs):
        for key, value in kwargs.items():
            self._ctx.set_option(key, value)

    def get_memory_usage(self):
        """Device kB held by the context (the RSS of the reference says nothing here)."""
        return 0.0 if self._ctx is None else self._ctx.device_bytes / 1024

    # HIP-event timings of the last query() (device side) and the source slice this rank owns
    @property
    def device_kernel_ms(self):
        return self._ctx.last_kernel_ms

    @property
    def device_total_ms(self):
        return self._ctx.last_total_ms

    @property
    def device_kernel(self):
        return self._ctx.last_kernel_name

    @property
    def shard(self):
        return self._shard

    def get_additional(self):
        if self._ctx is None:
            return {}
        return {
            "device_kernel_ms": self._ctx.last_kernel_ms,
            "device_total_ms": self._ctx.last_total_ms,
            "device_kernel": self._ctx.last_kernel_name,
            "n_gpus": 1 if self.comm is None else self.comm.world,
            "device_bytes": self._ctx.device_bytes,
            # what RCCL itself saw, and the all-reduce's share of device_total_ms (0 on one GPU)
            "rccl_ranks": self._ctx.rccl_ranks,
            "allreduce_ms": self._ctx.last_allreduce_ms,
            # "" or why a faster form was not taken (too few points per grid cell, the radius rule, ...)
            "dispatch_note": self._ctx.last_dispatch_note or getattr(self, "_dot_note", ""),
        }

    def done(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def __del__(self):
        # the runner calls done() only on the best instance (runner.py:174-176)
        try:
            self.done()
        except Exception:
            pass


class MI355XSolver(BaseSolver):
    """Solves K b = a with the HIP product as the operator: conjugate gradients for the positive
    definite Gaussian / exp(-r) matrices, MINRES for the symmetric indefinite inverse-distance
    matrix (zero diagonal, bruteforce.py:13-14)."""

    def __init__(self, *, kernel, dimension, normalize_rows=False, precision=np.float64,
                 device=0, rtol=1e-6, maxit=10000, comm=None, refine=None, inner_rtol=1e-3):
        super().__init__(kernel=kernel, dimension=dimension, normalize_rows=normalize_rows,
                         precision=precision)
        if kernel not in SUPPORTED_KERNELS:
            raise NotImplementedError(f"MI355XSolver doesn't support kernel {kernel}.")
        # refine="float32" (float64 solves only, an extension: the reference has one dense lstsq): mixed-precision
        # iterative refinement -- the residual a - K x is formed with the FLOAT64 operator, the correction K d = r is
        # solved to `inner_rtol` by CG on the FLOAT32 operator (the matrix-core cell form: ~10x cheaper per product),
        # x += d, until the float64 residual meets rtol.  The answer is judged exactly like the plain float64 solve.
        if refine not in (None, "float32"):
            raise ValueError("refine must be None or 'float32'")
        if refine and (np.dtype(precision) != np.float64 or kernel == "inverse-distance" or comm is not None):
            raise NotImplementedError("refine='float32' is for single-GPU float64 CG solves (gaussian, absolute-exponential, exp-dot)")
        self.refine = refine
        self.inner_rtol = float(inner_rtol)
        self._ctx32 = None
        self.outer_iterations = 0
        self.comm = comm  # sharding.Communicator: operator sharded over the sources, vectors replicated
        self._dtype_code, self._host_dtype = _lib.dtype_code(precision)
        if self._dtype_code == _lib.KMVP_BF16:
            raise NotImplementedError("MI355XSolver needs float16, float32 or float64")
        self._round = _lib.input_rounding(precision)  # float16: rounded inputs, float32 operator (scipy promotes too)
        self.device = device
        self.rtol = rtol
        self.maxit = maxit
        self._ctx = None
        self.iterations = 0
        self.residual = float("nan")
        self.converged = False
        self._dot = kernel == "exp-dot"  # K = D G D with D = diag(exp(|x|^2/2)), G the Gaussian matrix of the points / sqrt(2)
        self._device_kernel_fn = "gaussian" if self._dot else kernel
        self.method = "minres" if kernel == "inverse-distance" else "cg"
        self.name = (f"MI355XSolver({_precision_name(precision)}, {self.method}, rtol={rtol:g})" if not refine else
                     f"MI355XSolver({_precision_name(precision)}, {self.method} + refinement on {refine}, rtol={rtol:g})")

    def _cast(self, a):
        if self._round is not None:
            a = np.asarray(a, dtype=self._round)
        return np.ascontiguousarray(a, dtype=self._host_dtype)

    def prepare_data(self, *, source_points):
        self._prepare_data(source_points)
        if self.refine:
            # the same points (after the exp-dot scaling) as float32 in a second context: the inner operator
            if self._ctx32 is None:
                self._ctx32 = _lib.Context(self.device)
            self._ctx32.set_points(np.ascontiguousarray(self._y_device, dtype=np.float32), None, _lib.KMVP_F32)

    def _prepare_data(self, source_points):
        if self._dot:
            ys = np.asarray(self._cast(source_points), dtype=np.float64)
            source_points = ys * SQRT_HALF
            self._hx = _sq_norms_on_device(self._cast(source_points), False)
        y = self._cast(source_points)
        self._y_device = y
        self.M, self.D = y.shape
        if self._ctx is None:
            self._ctx = _lib.Context(self.device)
        world = 1 if self.comm is None else self.comm.world
        if world > 1:
            # every rank iterates on the full (replicated) Krylov vectors; the operator
            # is sharded over the sources and summed by the product's own all-reduce
            self._shard = sharding.shard_range(self.M, self.comm.rank, world)
            lo, hi = self._shard
            # (the sources keep the caller's order here: a rank's signal is a slice of the replicated Krylov vector)
            self.comm.attach(self._ctx)
            self._ctx.set_option("same_points_global", 1)
            self._ctx.set_points(np.ascontiguousarray(y[lo:hi]), y, self._dtype_code, j_offset=lo, M_total=self.M)
        else:
            self._shard = (0, self.M)
            self._ctx.set_points(y, None, self._dtype_code)

    def fit(self):
        """Nothing to factorise; the cell order of the float64 Gaussian operator is built here (kmvp_fit)."""
        self._ctx.fit(self._device_kernel_fn)
        if self._ctx32 is not None:
            self._ctx32.fit(self._device_kernel_fn)

    def prepare_query(self, *, target_signal):
        if self._dot:  # G (D b) = D^-1 a
            target_signal = np.asarray(target_signal, dtype=np.float64).reshape(self.M, -1) * np.exp(-self._hx).reshape(-1, 1)
        a = self._cast(target_signal)
        self._a = a.reshape(-1, 1) if a.ndim == 1 else a
        if self._a.ndim != 2 or self._a.shape[0] != self.M:
            raise ValueError(f"target_signal has shape {a.shape}, expected ({self.M}, E)")

    def set_query_arguments(self, rtol=None, maxit=None):
        if rtol is not None:
            self.rtol = rtol
        if maxit is not None:
            self.maxit = maxit

    def query(self):
        if self.refine:
            self._query_refined()
        else:
            self.res, self.iterations, self.residual, self.converged = self._ctx.cg_solve(
                self._device_kernel_fn, self._a, self.rtol, self.maxit)
        if self._dot:
            # The iteration ran on the SCALED system G z = D^-1 a (z = D b, D = diag(exp(|x|^2/2))); where D varies a lot
            # its residual says little about K b = a.  The verdict is therefore taken on the unscaled system, from one
            # more product:  a - K b = D (D^-1 a - G z).
            z = np.ascontiguousarray(self.res, dtype=self._host_dtype)
            self._ctx.set_signal(z)
            self._ctx.run(self._device_kernel_fn, False)
            r_scaled = np.asarray(self._a, dtype=np.float64) - self._ctx.get_result(self.M, z.shape[1])
            with np.errstate(over="ignore", invalid="ignore"):
                d = np.exp(self._hx).reshape(-1, 1)
                num = np.linalg.norm(d * r_scaled, axis=0)
                den = np.linalg.norm(d * np.asarray(self._a, dtype=np.float64), axis=0)
                den[den == 0] = 1.0
                self.scaled_residual = self.residual
                self.residual = float(np.max(num / den))
            self.converged = bool(np.isfinite(self.residual) and self.residual <= 1.5 * self.rtol)
            self.res = self.res * np.exp(-self._hx).reshape(-1, 1)  # b = D^-1 (D b)

    def _query_refined(self):
        """Mixed-precision iterative refinement (see __init__).  Every outer step: float64 residual (one product of
        the float64 context), float32 CG on the scaled residual, float64 update."""
        a = np.asarray(self._a, dtype=np.float64)
        anorm = np.linalg.norm(a, axis=0)
        anorm[anorm == 0] = 1.0
        x = np.zeros_like(a)
        r = a.copy()
        self.iterations, self.outer_iterations, self.converged = 0, 0, False
        rel = float(np.max(np.linalg.norm(r, axis=0) / anorm))
        best = rel
        # why the outer loop ended (get_additional "refinement_stop_reason"): tolerance | stagnation | maxit |
        # outer-limit | non-finite; with the last inner solve's own verdict beside it, so that a stalled refinement
        # (inner solve converged, outer residual stuck) can be told from an exhausted iteration budget
        self.stop_reason, self.inner_residual, self.inner_converged = "outer-limit", float("nan"), None
        for _ in range(40):
            if rel <= self.rtol:
                self.stop_reason = "tolerance"
                break
            if self.iterations >= self.maxit:
                self.stop_reason = "maxit"
                break
            scale = np.max(np.abs(r), axis=0)
            scale[scale == 0] = 1.0  # (a power-of-two free scaling is not needed: float32 has the range, this keeps it centred)
            d, iters, inner_res, inner_ok = self._ctx32.cg_solve(
                self._device_kernel_fn, np.ascontiguousarray(r / scale, dtype=np.float32), self.inner_rtol,
                max(1, self.maxit - self.iterations))
            self.iterations += iters
            self.outer_iterations += 1
            self.inner_residual, self.inner_converged = float(inner_res), bool(inner_ok)
            x_new = x + d * scale
            self._ctx.set_signal(x_new)
            self._ctx.run(self._device_kernel_fn, False)
            r_new = a - self._ctx.get_result(self.M, a.shape[1])
            rel_new = float(np.max(np.linalg.norm(r_new, axis=0) / anorm))
            if not np.isfinite(rel_new):
                self.stop_reason = "non-finite"
                break
            if rel_new > 0.9 * best:
                # the float32 correction did not help (inner tolerance beyond what a float32 operator can reach on
                # this matrix): keep the last good iterate and say so through `converged`
                self.stop_reason = "stagnation"
                break
            x, r, rel = x_new, r_new, rel_new
            best = rel
        else:
            if rel <= self.rtol:
                self.stop_reason = "tolerance"
        self.res, self.residual = x, rel
        self.converged = bool(np.isfinite(rel) and rel <= 1.5 * self.rtol)

    def get_memory_usage(self):
        return 0.0 if self._ctx is None else self._ctx.device_bytes / 1024

    def get_additional(self):
        extra = {"cg_iterations": self.iterations, "cg_relative_residual": self.residual,
                 "cg_converged": bool(self.converged), "n_gpus": 1 if self.comm is None else self.comm.world}
        if self._ctx is not None:
            extra["device_kernel"] = self._ctx.last_kernel_name  # the operator's pair-loop kernel
            extra["rccl_ranks"] = self._ctx.rccl_ranks
        if self._dot and hasattr(self, "scaled_residual"):
            extra["cg_scaled_system_residual"] = self.scaled_residual  # of G (D b) = D^-1 a, what the iteration saw
        if self.refine and self._ctx32 is not None:
            extra["refinement_steps"] = self.outer_iterations
            extra["inner_device_kernel"] = self._ctx32.last_kernel_name
            extra["refinement_stop_reason"] = getattr(self, "stop_reason", "")
            extra["refinement_last_inner_residual"] = getattr(self, "inner_residual", float("nan"))
            extra["refinement_last_inner_converged"] = bool(getattr(self, "inner_converged", False))
        return extra

    def done(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None
        if getattr(self, "_ctx32", None) is not None:
            self._ctx32.close()
            self._ctx32 = None

    def __del__(self):
        try:
            self.done()
        except Exception:
            pass
