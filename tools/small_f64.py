"""float64 Gaussian products at the reference's dataset sizes (1000 ... 10000 sphere points) against the segment count: time and error."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kmvp_oracle
from kernel_matrix_benchmarks_amd import _lib
for n in (1000, 2000, 10000):
    y = kmvp_oracle.uniform_sphere_points(n); b = np.random.RandomState(n).randn(n, 1)
    want = kmvp_oracle.product(kernel="gaussian", source_points=y, source_signal=b)
    for seg in (0, 4, 8, 16, 32, 64):
        ctx = _lib.Context(0)
        if seg: ctx.set_option("segments", seg)
        ctx.set_points(y, None, _lib.KMVP_F64); ctx.set_signal(b)
        ctx.run("gaussian", False); ctx.run("gaussian", False)
        ks = []
        for _ in range(10):
            ctx.run("gaussian", False); ks.append(ctx.last_kernel_ms)
        a = ctx.get_result(n, 1)
        print(f"f64 gaussian n={n} segments={seg}: kernel {np.median(ks)*1e3:.1f} us err {np.max(np.abs(a-want))/np.max(np.abs(want)):.1e}", flush=True)
        ctx.close()
