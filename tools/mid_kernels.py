"""lowd_mid_kernel: kernels / precisions / signal widths the matrix-core paths do not take (N = M = 1e5;
last line: the config-3 shape in float32)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib
def run(n, D, E, kernel, norm, code):
    rs = np.random.RandomState(1)
    npdt = np.float64 if code == _lib.KMVP_F64 else np.float32
    y = (rs.rand(n, D) / np.sqrt(D)).astype(npdt); b = rs.randn(n, E).astype(npdt)
    ctx = _lib.Context(0)
    ctx.set_points(y, None, code); ctx.set_signal(b)
    ctx.run(kernel, norm); ctx.run(kernel, norm)
    ms = []
    for _ in range(3):
        ctx.run(kernel, norm); ms.append(ctx.last_kernel_ms)
    print(f"n={n} D={D:3d} E={E:2d} {kernel:20s} norm={norm!s:5s} {'f64' if code == _lib.KMVP_F64 else 'f32'}: {min(ms):9.2f} ms  {n*n/(min(ms)*1e-3):.2e} pairs/s  {ctx.last_kernel_name}", flush=True)
    ctx.close()
for D in (10, 16, 32):
    run(100000, D, 1, "absolute-exponential", False, _lib.KMVP_F32)
    run(100000, D, 1, "inverse-distance", False, _lib.KMVP_F32)
    run(100000, D, 1, "gaussian", False, _lib.KMVP_F64)
run(65536, 64, 64, "absolute-exponential", True, _lib.KMVP_F32)
