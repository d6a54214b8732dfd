"""Low D, many signal columns (attention with E value channels) at 1e5 points: the matrix-core forms (auto) against the
column-blocked difference form (fast_sqdists = 0), per kernel function; targets == sources and targets != sources (the
per-target online shift of the f16-split products).  usage: python tools/high_e.py [kernel ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib
n = 100000
rs = np.random.RandomState(1)
y = rs.rand(n, 3).astype(np.float32)
x = rs.rand(n, 3).astype(np.float32)
kernels = sys.argv[1:] or ["gaussian", "absolute-exponential", "inverse-distance"]
for kernel in kernels:
    for E in (4, 16, 64):
        b = rs.randn(n, E).astype(np.float32)
        for norm in (True,):
            for label, targets, fast in (("x == y auto", None, -1), ("x != y auto", x, -1), ("x == y difference form", None, 0)):
                ctx = _lib.Context(0)
                ctx.set_option("fast_sqdists", fast)
                ctx.set_points(y, targets, _lib.KMVP_F32); ctx.set_signal(b)
                for _ in range(20): ctx.run(kernel, norm)
                ms = []
                for _ in range(5):
                    ctx.run(kernel, norm); ms.append(ctx.last_kernel_ms)
                print(f"{kernel:22s} D=3 E={E:2d} norm={norm!s:5s} {label:24s}: {min(ms):8.2f} ms  {n*n/(min(ms)*1e-3):.2e} pairs/s  "
                      f"{ctx.last_kernel_name} {ctx.last_dispatch_note[:40]}", flush=True)
                ctx.close()
