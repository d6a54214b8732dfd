"""fast_kernel target tiles per wave (fast_tiles 1 / 2 / 4) at N = M = 1e6, Gaussian."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib
n = 1_000_000
rs = np.random.RandomState(n + 3)
y = rs.rand(n, 3).astype(np.float32); b = rs.randn(n, 1).astype(np.float32)
for tiles in (1, 2, 4):
    ctx = _lib.Context(0)
    ctx.set_option("fast_sqdists", 1); ctx.set_option("fast_tiles", tiles)
    ctx.set_points(y, None, _lib.KMVP_F32); ctx.set_signal(b)
    ctx.run("gaussian", False)
    ms = []
    for _ in range(4):
        ctx.run("gaussian", False); ms.append(ctx.last_kernel_ms)
    print(f"fast_kernel tiles={tiles}: {min(ms):.2f} ms", flush=True)
    ctx.close()
for T, feed in ((1, 1), (2, 1), (1, 0), (2, 0)):
    ctx = _lib.Context(0)
    ctx.set_option("fast_sqdists", 0); ctx.set_option("targets_per_lane", T); ctx.set_option("feed", feed)
    ctx.set_points(y, None, _lib.KMVP_F32); ctx.set_signal(b)
    ctx.run("gaussian", False)
    ms = []
    for _ in range(3):
        ctx.run("gaussian", False); ms.append(ctx.last_kernel_ms)
    print(f"lowd_kernel T={T} feed={feed}: {min(ms):.2f} ms", flush=True)
    ctx.close()
