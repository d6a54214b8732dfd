"""Times cell_kernel builds (libkmvp variants given as paths) at the headline shape: python tools/cell_variants.py lib.so tiles [tiles...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib
if sys.argv[1] != "default":
    _lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])
n = 1000000
rs = np.random.RandomState(n + 3)
y = rs.rand(n, 3).astype(np.float32); b = rs.randn(n, 1).astype(np.float32)
ref = None
for tt in [int(t) for t in sys.argv[2:]]:
    ctx = _lib.Context(0)
    ctx.set_option("fast_sqdists", 3); ctx.set_option("fast_tiles", tt)
    ctx.set_points(y, None, _lib.KMVP_F32); ctx.set_signal(b)
    ctx.run("gaussian", False); ctx.run("gaussian", False)
    ms = []
    for _ in range(4):
        ctx.run("gaussian", False); ms.append(ctx.last_kernel_ms)
    out = ctx.get_result(n, 1)
    if ref is None: ref = out
    print(f"{sys.argv[1]} TT={tt}: {ctx.last_kernel_name} {min(ms):.2f} ms  {n*n/(min(ms)*1e-3):.3e} pairs/s  diff vs first {np.max(np.abs(out-ref))/np.max(np.abs(ref)):.1e}", flush=True)
    ctx.close()
