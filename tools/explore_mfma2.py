"""C3 timing of mfma_kernel variants (target tiles per wave x segments)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd import _lib
import c_oracle
def rel(a, b): return np.max(np.sqrt(np.sum((a-b)**2, -1))) / np.max(np.sqrt(np.sum(b**2, -1)))
ctx = _lib.Context(0)
n, D, E = 65536, 64, 64
rs = np.random.RandomState(n + D)
y = (rs.rand(n, D) / np.sqrt(D)); b = rs.randn(n, E)
rows = np.random.RandomState(0).choice(n, 128, replace=False)
for kernel in ("absolute-exponential", "gaussian"):
    want = c_oracle.product(kernel=kernel, source_points=y, source_signal=b, normalize_rows=True, rows=rows)
    ctx.set_points(y.astype(np.float32), None, _lib.KMVP_BF16); ctx.set_signal(b.astype(np.float32))
    for tw in (1, 2):
        for seg in (0, 4, 8, 16):
            ctx.set_option("targets_per_lane", tw); ctx.set_option("segments", seg)
            ctx.run(kernel, True)
            ms = []
            for _ in range(7):
                ctx.run(kernel, True); ms.append(ctx.last_kernel_ms)
            got = ctx.get_result(n, E)
            print(f"C3 {kernel:22s} TW={tw} seg={seg:2d} kernel_ms min {min(ms):.3f} med {sorted(ms)[3]:.3f} pairs/s={n*n/(min(ms)*1e-3):.3e} rel={rel(got[rows], want):.2e}", flush=True)
