"""CPU oracle for the kernel matrix-vector product path -- TEST INFRASTRUCTURE ONLY.

This module is a numpy restatement of the arithmetic of the reference's
bruteforce plugin.  It exists to CHECK the HIP path; it is never the thing that
is shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``kernel_matrix_benchmarks_amd``) must not, and does not, import anything from
``oracle/``.

Parity status: PINNED.  Every function below is compared with the imported
reference (``/root/reference/kernel_matrix_benchmarks/algorithms/bruteforce.py``)
by ``tests/make_golden.py`` when the fixtures under ``tests/golden/`` are
generated, and against those committed fixtures by ``tests/test_oracle.py``.

Reference lines restated here (all in ``kernel_matrix_benchmarks/algorithms/``):

* ``bruteforce.py:8-15``   inverse_square_root: ``1/sqrt(max(s,0))`` then the flat
  indices ``::M+1`` of the (N,M) matrix are set to 0
* ``bruteforce.py:18-22``  kernel_functions
* ``bruteforce.py:25-58``  kernel_matrix (slow ``(N,M,D)`` difference form and the
  fast ``|x|^2+|y|^2-2xy`` BLAS form)
* ``bruteforce.py:130-153`` BruteForceProductBLAS.query (four branches)
* ``bruteforce.py:205-207`` BruteForceSolverLAPACK.query (``lstsq``)
* ``base.py:107-116``      get_result -> float64 C-contiguous

Unlike the reference this restatement is ROW-BLOCKED: it never holds more than
``block_rows x M`` kernel values, so it also serves as the checker for row
subsets of the 1e6 / 1e7 point configurations the dense reference cannot run.
"""
import numpy as np

KERNELS = ("gaussian", "absolute-exponential", "inverse-distance")


def zero_column(i, M):
    """Column of row ``i`` that ``inverse_square_root`` zeroes, or -1.

    bruteforce.py:13-14 zeroes flat indices ``k*(M+1)`` of the row-major (N,M)
    buffer.  Flat index ``i*M + j`` is a multiple of ``M+1`` iff
    ``j == i mod (M+1)`` (because ``M == -1 mod (M+1)``); that column exists
    only when it is ``< M``.  For ``N <= M`` this is the diagonal ``j == i``; for
    ``N > M`` it wraps (row ``M`` has none, row ``M+1`` has column 0, ...).
    """
    i = np.asarray(i, dtype=np.int64)
    jz = i % (M + 1)
    return np.where(jz < M, jz, -1)


def sqdists_block(x, y, fast_sqdists):
    """Squared distances of a block of targets ``x`` (n,D) to all sources ``y`` (M,D).

    bruteforce.py:36-54.  Computed in the dtype of the inputs.
    """
    n, D = x.shape
    M = y.shape[0]
    if fast_sqdists:
        ysq = (y ** 2).sum(-1)
        xsq = (x ** 2).sum(-1)
        return xsq.reshape(n, 1) + ysq.reshape(1, M) - 2 * x @ y.T
    diffs = x.reshape(n, 1, D) - y.reshape(1, M, D)
    return np.sum(diffs ** 2, axis=-1)


def kernel_block(kernel, s, rows, M, j_offset=0, M_total=None):
    """Kernel values for a block of squared distances ``s`` (n, M_local).

    ``rows`` are the GLOBAL target indices of the block's rows; ``j_offset`` is
    the global index of the block's first column and ``M_total`` the global
    number of sources (they differ from 0 / M only when the sources are
    sharded), so the inverse-distance zero pattern follows the global flat
    index rule of bruteforce.py:13-14.
    """
    if kernel == "gaussian":
        return np.exp(-s)
    if kernel == "absolute-exponential":
        return np.exp(-np.sqrt(np.maximum(s, 0)))
    if kernel == "inverse-distance":
        with np.errstate(divide="ignore"):
            k = 1 / np.sqrt(np.maximum(s, 0))
        Mt = M if M_total is None else M_total
        jz = zero_column(rows, Mt) - j_offset
        hit = (jz >= 0) & (jz < s.shape[1])
        k[np.nonzero(hit)[0], jz[hit]] = 0
        return k
    raise NotImplementedError(f"unknown kernel {kernel}")


def product(
    *,
    kernel,
    source_points,
    target_points=None,
    source_signal=None,
    normalize_rows=False,
    density_estimation=False,
    precision=np.float64,
    fast_sqdists=False,
    rows=None,
    block_rows=None,
    j_offset=0,
    M_total=None,
    raw_sums=False,
):
    """a_i = sum_j k(x_i, y_j) b_j for the target rows ``rows`` (default: all).

    Mirrors prepare_data / fit / prepare_query / query / get_result of
    BruteForceProductBLAS (bruteforce.py:89-153, base.py:107-116): inputs are
    cast to ``precision``, the arithmetic runs in ``precision``, the result is
    returned as float64.  ``target_points=None`` means ``same_points``.

    ``raw_sums=True`` returns the un-normalised numerator and the denominator
    (n,E) and (n,1) -- what one source shard contributes before the all-reduce.
    """
    precision = np.dtype(precision)
    y = np.ascontiguousarray(source_points, dtype=precision)
    x = y if target_points is None else np.ascontiguousarray(target_points, dtype=precision)
    M, D = y.shape
    N = x.shape[0]
    if density_estimation or source_signal is None:
        b = None
        E = 1
    else:
        b = np.ascontiguousarray(source_signal, dtype=precision)
        E = b.shape[1]
    rows = np.arange(N, dtype=np.int64) if rows is None else np.asarray(rows, dtype=np.int64)
    n = rows.shape[0]

    if normalize_rows and b is None and not raw_sums:
        # bruteforce.py:134-138: rows of a normalised matrix sum to one.
        return np.ones((n, 1), dtype=np.float64)

    if block_rows is None:
        block_rows = max(1, min(n, int(2 ** 25 // max(1, M * (1 if fast_sqdists else D)))))
    num = np.empty((n, E), dtype=precision)
    den = np.empty((n, 1), dtype=precision)
    for r0 in range(0, n, block_rows):
        rr = rows[r0 : r0 + block_rows]
        s = sqdists_block(x[rr], y, fast_sqdists)
        K = kernel_block(kernel, s, rr, M, j_offset=j_offset, M_total=M_total)
        if b is None:
            # bruteforce.py:150  K.sum(-1, keepdims=True)
            num[r0 : r0 + block_rows] = np.sum(K, -1, keepdims=True)
            den[r0 : r0 + block_rows] = num[r0 : r0 + block_rows]
        elif normalize_rows or raw_sums:
            # bruteforce.py:142-145  K @ [b | 1]
            sig1 = np.concatenate((b, np.ones_like(b[..., :1])), axis=1)
            rs = K @ sig1
            num[r0 : r0 + block_rows] = rs[..., :-1]
            den[r0 : r0 + block_rows] = rs[..., -1:]
        else:
            # bruteforce.py:153  K @ b
            num[r0 : r0 + block_rows] = K @ b
    if raw_sums:
        return num.astype(np.float64), den.astype(np.float64)
    res = num / den if normalize_rows else num
    return np.ascontiguousarray(res, dtype=np.float64)


def exp_dot_product(*, source_points, target_points=None, source_signal=None, normalize_rows=False, block_rows=2048):
    """a_i = sum_j exp(<x_i, y_j>) b_j [/ sum_j exp(<x_i, y_j>)] in float64, evaluated directly.

    PARITY UNPINNED: the kernel is defined only in the reference's README (README.md:51-59, "an exponential kernel
    k(x_i, y_j) = exp(<x_i, y_j>)" for attention layers); none of its plugins implements it (bruteforce.py:18-22),
    so there is no reference output to pin it.  This direct evaluation (row max subtracted before exp, the textbook
    softmax stabilisation) checks the Gaussian identity the plugin uses."""
    y = np.asarray(source_points, dtype=np.float64)
    x = y if target_points is None else np.asarray(target_points, dtype=np.float64)
    b = np.ones((y.shape[0], 1)) if source_signal is None else np.asarray(source_signal, dtype=np.float64)
    out = np.empty((x.shape[0], b.shape[1]))
    for r0 in range(0, x.shape[0], block_rows):
        s = x[r0 : r0 + block_rows] @ y.T
        m = s.max(axis=1, keepdims=True)
        p = np.exp(s - m)
        num = p @ b
        out[r0 : r0 + block_rows] = num / p.sum(axis=1, keepdims=True) if normalize_rows else num * np.exp(m)
    return out


def kernel_matrix(*, kernel, source_points, target_points=None, fast_sqdists=False):
    """Dense K (N,M), bruteforce.py:25-58.  Small shapes only."""
    y = source_points
    x = y if target_points is None else target_points
    s = sqdists_block(x, y, fast_sqdists)
    return kernel_block(kernel, s, np.arange(x.shape[0]), y.shape[0])


def solve(*, kernel, source_points, target_signal, precision=np.float64, fast_sqdists=False):
    """b = lstsq(K, a)[0], bruteforce.py:193-207 (LAPACK gelsd, minimum norm)."""
    from scipy.linalg import lstsq

    precision = np.dtype(precision)
    y = np.ascontiguousarray(source_points, dtype=precision)
    a = np.ascontiguousarray(target_signal, dtype=precision)
    K = kernel_matrix(kernel=kernel, source_points=y, fast_sqdists=fast_sqdists)
    return np.ascontiguousarray(lstsq(K, a)[0], dtype=np.float64)


def relative_residual(*, kernel, source_points, solution, target_signal, rows=None):
    """||K b - a|| / ||a|| in float64 on ``rows`` -- the solver's success measure
    (the Gaussian matrix is numerically singular, so the harness'
    ``result - source_signal`` is not a meaningful parity measure)."""
    a = np.asarray(target_signal, dtype=np.float64)
    Kb = product(
        kernel=kernel, source_points=source_points, source_signal=solution, rows=rows
    )
    ar = a if rows is None else a[np.asarray(rows)]
    return float(np.linalg.norm(Kb - ar) / np.linalg.norm(ar))


def uniform_cube(n_points, dimension, radius=1, E=1):
    """Inputs of the reference's ``uniform_cube`` generator, datasets.py:256-266:
    ``seed(n+D)``, ``rand(n,D)*radius``, ``randn(n,1)`` (E>1 draws ``randn(n,E)``,
    identical to the reference for E=1)."""
    rs = np.random.RandomState(n_points + dimension)  # == numpy.random.seed(n+D) stream
    y = radius * rs.rand(n_points, dimension)
    b = rs.randn(n_points, E)
    return y, b


def uniform_sphere_points(n_points, radius=1):
    """Golden-angle spiral of datasets.py:210-225 (D=3 only; signal is unseeded there)."""
    import math

    pts = np.zeros((n_points, 3))
    phi = math.pi * (3.0 - math.sqrt(5.0))
    for i in range(n_points):
        yy = 1 - (i / float(n_points - 1)) * 2
        ry = math.sqrt(1 - yy * yy)
        theta = phi * i
        pts[i, 0] = radius * math.cos(theta) * ry
        pts[i, 1] = radius * yy
        pts[i, 2] = radius * math.sin(theta) * ry
    return pts


def result_errors(error):
    """max / mean / median / rmse of the per-row L2 norms, plotting/metrics.py:53-59."""
    norms = np.sqrt(np.sum(np.asarray(error, dtype=np.float64) ** 2, axis=-1))
    return {
        "max": float(np.max(norms)),
        "mean": float(np.mean(norms)),
        "median": float(np.median(norms)),
        "rmse": float(np.sqrt(np.mean(norms ** 2))),
    }
