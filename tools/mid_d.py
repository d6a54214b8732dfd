"""Point dimension sweep at E = 1, N = M = 1e5, float32 Gaussian: where the specialised kernels end."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib
n = 100000
rs = np.random.RandomState(1)
for D in (3, 8, 16, 39, 40, 64, 100, 128, 129, 160, 256, 512):
    y = (rs.rand(n, D) / np.sqrt(D)).astype(np.float32); b = rs.randn(n, 1).astype(np.float32)
    for prec, code in (("f32", _lib.KMVP_F32), ("bf16", _lib.KMVP_BF16)):
        if prec == "bf16" and (D < 16 or D > 128): continue
        ctx = _lib.Context(0)
        ctx.set_points(y, None, code); ctx.set_signal(b)
        ctx.run("gaussian", False); ctx.run("gaussian", False)
        ms = []
        for _ in range(3):
            ctx.run("gaussian", False); ms.append(ctx.last_kernel_ms)
        print(f"D={D:3d} {prec}: {min(ms):8.2f} ms  {n*n/(min(ms)*1e-3):.2e} pairs/s  {ctx.last_kernel_name}", flush=True)
        ctx.close()
