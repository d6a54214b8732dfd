"""cell_kernel at the headline shape against the number of source segments (auto = L2-sized segments)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib
n = 1000000
rs = np.random.RandomState(n + 3)
y = rs.rand(n, 3).astype(np.float32); b = rs.randn(n, 1).astype(np.float32)
for seg in (0, 8, 16, 24, 32, 48, 64, 96):
    ctx = _lib.Context(0)
    if seg: ctx.set_option("segments", seg)
    ctx.set_points(y, None, _lib.KMVP_F32); ctx.set_signal(b)
    ctx.run("gaussian", False); ctx.run("gaussian", False)
    ms = []
    for _ in range(4):
        ctx.run("gaussian", False); ms.append(ctx.last_kernel_ms)
    print(f"segments {seg or 'auto':>4}: {ctx.last_kernel_name} {min(ms):.2f} ms (step {ctx.last_total_ms:.2f} ms)", flush=True)
    ctx.close()
