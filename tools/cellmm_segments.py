"""cellmm_kernel at N = M = 1e6 against the number of source segments (option "segments"): kernel time per setting."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from kernel_matrix_benchmarks_amd import _lib
n = 1_000_000
rs = np.random.RandomState(n + 3)
y = rs.rand(n, 3).astype(np.float32); b = rs.randn(n, 1).astype(np.float32)
ctx = _lib.Context(0); ctx.set_points(y, None, _lib.KMVP_F32); ctx.set_signal(b)
for seg in (0, 4, 8, 16, 24, 32, 0):
    ctx.set_option("segments", seg)
    ctx.run("gaussian", False)
    ms = []
    for _ in range(8):
        ctx.run("gaussian", False); ms.append(ctx.last_kernel_ms)
    print(f"segments {seg:2d}: {ctx.last_kernel_name} min {min(ms):.3f} mean {np.mean(ms):.3f} total {ctx.last_total_ms:.3f} device MB {ctx.device_bytes/1e6:.0f}", flush=True)
ctx.close()
