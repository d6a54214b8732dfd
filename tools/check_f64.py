"""float64 products (Gaussian, exp(-r)) on clouds of growing extent against the numpy oracle, and the float64 cell form at 1e5 points."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd import _lib
import kmvp_oracle
ctx = _lib.Context(0)
for scale in (1.0, 5.0, 30.0):
    rs = np.random.RandomState(5); y = rs.rand(700, 3) * scale; x = rs.rand(500, 3) * scale; b = rs.randn(700, 2)
    for kernel in ("gaussian", "absolute-exponential"):
        want = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b)
        ctx.set_points(y, x, _lib.KMVP_F64); ctx.set_signal(b); ctx.run(kernel, False)
        got = ctx.get_result(500, 2)
        print(kernel, "scale", scale, "rel", np.max(np.abs(got - want)) / np.max(np.abs(want)), "max elementwise rel", np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)))
n = 100000
rs = np.random.RandomState(n + 3); y = rs.rand(n, 3); b = rs.randn(n, 1)
ctx.set_points(y, None, _lib.KMVP_F64); ctx.set_signal(b)
for k in ("gaussian", "absolute-exponential", "inverse-distance"):
    ctx.run(k, False); ms = []
    for _ in range(5):
        ctx.run(k, False); ms.append(ctx.last_kernel_ms)
    print(k, "f64 1e5 kernel_ms", min(ms), "pairs/s %.3e" % (n * n / (min(ms) * 1e-3)))
