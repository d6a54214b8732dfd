"""Low D, many signal columns (attention with E value channels): generic kernel vs E <= 4 specialised."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib
n = 100000
rs = np.random.RandomState(1)
y = rs.rand(n, 3).astype(np.float32)
for E in (1, 4, 5, 8, 16, 64):
    b = rs.randn(n, E).astype(np.float32)
    for norm in (False, True):
        ctx = _lib.Context(0)
        ctx.set_points(y, None, _lib.KMVP_F32); ctx.set_signal(b)
        ctx.run("gaussian", norm); ctx.run("gaussian", norm)
        ms = []
        for _ in range(3):
            ctx.run("gaussian", norm); ms.append(ctx.last_kernel_ms)
        print(f"D=3 E={E:2d} norm={norm!s:5s}: {min(ms):8.2f} ms  {n*n/(min(ms)*1e-3):.2e} pairs/s  {ctx.last_kernel_name}", flush=True)
        ctx.close()
