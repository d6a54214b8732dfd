"""One workload for tools/profile_pmc.sh: argv[1] = f32 (cell_kernel at 1e6) or f64 (cell64_kernel / lowd_kernel at 2e5)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib
which = sys.argv[1]
n = 1000000 if which == "f32" else 200000
rs = np.random.RandomState(n + 3)
y = rs.rand(n, 3); b = rs.randn(n, 1)
if os.environ.get("KMVP_ZERO_B"):
    b[:] = 0
for fast in ((3,) if which == "f32" else (3, 0)):
    ctx = _lib.Context(0)
    ctx.set_option("fast_sqdists", fast)
    if which == "f32":
        ctx.set_points(y.astype(np.float32), None, _lib.KMVP_F32); ctx.set_signal(b.astype(np.float32))
    else:
        ctx.set_points(y, None, _lib.KMVP_F64); ctx.set_signal(b)
    for _ in range(3):
        ctx.run("gaussian", False)
    print(ctx.last_kernel_name, ctx.last_kernel_ms)
    ctx.close()
