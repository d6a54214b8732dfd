"""float64 Gaussian: cell64_kernel against lowd_kernel and the oracle (uniform cube): kernel ms and errors."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from kernel_matrix_benchmarks_amd import _lib
import kmvp_oracle
for n in [int(float(v)) for v in sys.argv[1:]]:
    y, b = kmvp_oracle.uniform_cube(n, 3)
    rows = np.random.RandomState(1).choice(n, size=min(n, 300), replace=False)
    want = kmvp_oracle.product(kernel="gaussian", source_points=y, target_points=y[rows], source_signal=b)
    for name, fast in (("lowd", 0), ("cell64", 3), ("auto", -1)):
        ctx = _lib.Context(0)
        ctx.set_option("fast_sqdists", fast)
        ctx.set_points(y, None, _lib.KMVP_F64); ctx.set_signal(b)
        ctx.run("gaussian", False); ctx.run("gaussian", False)
        ms = []
        for _ in range(3):
            ctx.run("gaussian", False); ms.append(ctx.last_kernel_ms)
        out = ctx.get_result(n, 1)
        err = np.max(np.abs(out[rows] - want)) / np.max(np.abs(want))
        print(f"n={n} {name}: {ctx.last_kernel_name} kernel {min(ms):.3f} ms total {ctx.last_total_ms:.3f} ms  {n*n/(min(ms)*1e-3):.3e} pairs/s  rel_err {err:.2e}", flush=True)
        ctx.close()
