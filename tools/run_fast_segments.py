"""fast_kernel / cfast_kernel at N = M = n for a list of segment counts (L2 residency vs re-reads).
usage: python tools/run_fast_segments.py kernel n seg [seg ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib
kernel, n = sys.argv[1], int(float(sys.argv[2]))
rs = np.random.RandomState(n + 3)
y = rs.rand(n, 3).astype(np.float32); b = rs.randn(n, 1).astype(np.float32)
ctx = _lib.Context(0)
ctx.set_points(y, None, _lib.KMVP_F32); ctx.set_signal(b)
for seg in [int(s) for s in sys.argv[3:]]:
    ctx.set_option("segments", seg)
    ctx.run(kernel, False)
    ms = []
    for _ in range(4):
        ctx.run(kernel, False); ms.append(ctx.last_kernel_ms)
    print(f"{kernel} n={n} segments={seg} {ctx.last_kernel_name}: {min(ms):.2f} ms (total {ctx.last_total_ms:.2f}) device_MB {ctx.device_bytes/1e6:.0f}", flush=True)
ctx.close()
