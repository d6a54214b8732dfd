"""Centred split-bf16 path (cfast_kernel): accuracy vs the fp64 oracle, speed vs the difference form."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd import _lib
import kmvp_oracle, c_oracle

ctx = _lib.Context(0)
ctx.set_option("fast_sqdists", 2)
for (N, M, D, same) in ((300, 300, 3, True), (257, 193, 3, False), (193, 257, 2, False), (300, 300, 1, True), (130, 97, 4, False), (1000, 1000, 3, True)):
    rs = np.random.RandomState(N + D)
    y = rs.rand(M, D); x = None if same else rs.rand(N, D); b = rs.randn(M, 1)
    for kernel in ("gaussian", "absolute-exponential", "inverse-distance"):
        if kernel == "inverse-distance" and not same:
            continue
        for nr, de in ((False, False), (True, False), (False, True)):
            want = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b, normalize_rows=nr, density_estimation=de)
            ctx.set_points(y.astype(np.float32), None if same else x.astype(np.float32), _lib.KMVP_F32)
            ctx.set_signal(None if de else b.astype(np.float32))
            ctx.run(kernel, nr)
            got = ctx.get_result(N, 1)
            print(f"N={N} M={M} D={D} same={same!s:5s} {kernel:22s} nr={nr!s:5s} de={de!s:5s} {ctx.last_kernel_name} rel={np.max(np.abs(got-want))/np.max(np.abs(want)):.2e}", flush=True)

n = 1_000_000
rs = np.random.RandomState(n + 3)
y64 = rs.rand(n, 3); b64 = rs.randn(n, 1)
y = y64.astype(np.float32); b = b64.astype(np.float32)
rows = np.random.RandomState(0).choice(n, 512, replace=False)
ctx.set_points(y, None, _lib.KMVP_F32); ctx.set_signal(b)
for kernel in ("inverse-distance", "absolute-exponential", "gaussian"):
    want = c_oracle.product(kernel=kernel, source_points=y64, source_signal=b64, rows=rows)
    for fast, tt in ((0, 0), (2, 1), (2, 2), (2, 4)):
        ctx.set_option("fast_sqdists", fast); ctx.set_option("fast_tiles", tt)
        ctx.run(kernel, False)
        best = 1e9
        for _ in range(3):
            ctx.run(kernel, False); best = min(best, ctx.last_kernel_ms)
        got = ctx.get_result(n, 1)
        err = np.max(np.abs(got[rows] - want)) / np.max(np.abs(want))
        print(f"1e6 {kernel:22s} mode={fast} TT={tt} {ctx.last_kernel_name:12s} kernel_ms={best:8.3f} pairs/s={n*n/(best*1e-3):.3e} rel={err:.2e} finite={bool(np.isfinite(got).all())}", flush=True)
