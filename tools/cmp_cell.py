"""cell_kernel against fast_kernel and the fp64 oracle (Gaussian, uniform cube, float32): kernel ms and errors."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from kernel_matrix_benchmarks_amd import _lib
import kmvp_oracle
n = int(float(sys.argv[1]))
norm = len(sys.argv) > 2 and sys.argv[2] == "norm"
y, b = kmvp_oracle.uniform_cube(n, 3)
y32, b32 = y.astype(np.float32), b.astype(np.float32)
rows = np.random.RandomState(1).choice(n, size=min(n, 512), replace=False)
want = kmvp_oracle.product(kernel="gaussian", source_points=y32.astype(np.float64), target_points=y32[rows].astype(np.float64),
                           source_signal=b32.astype(np.float64), normalize_rows=norm)
res = {}
for name, fast, tt in (("fast4", 1, 4), ("cell1", 3, 1), ("cell2", 3, 2), ("cell4", 3, 4)):
    ctx = _lib.Context(0)
    ctx.set_option("fast_sqdists", fast)
    ctx.set_option("fast_tiles", tt)
    ctx.set_points(y32, None, _lib.KMVP_F32); ctx.set_signal(b32)
    import time
    t0 = time.perf_counter(); ctx.run("gaussian", norm); first_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); ctx.run("gaussian", norm); second_ms = (time.perf_counter() - t0) * 1e3
    ms = []
    for _ in range(3):
        ctx.run("gaussian", norm); ms.append(ctx.last_kernel_ms)
    out = ctx.get_result(n, 1)
    err = np.max(np.abs(out[rows] - want)) / np.max(np.abs(want))
    print(f"{name}: {ctx.last_kernel_name} kernel {min(ms):.2f} ms total {ctx.last_total_ms:.2f} ms  {n*n/(min(ms)*1e-3):.3e} pairs/s  rel_err {err:.2e}  dev_bytes {ctx.device_bytes/1e6:.0f} MB  first/second query wall {first_ms:.1f}/{second_ms:.1f} ms", flush=True)
    ctx.close()
