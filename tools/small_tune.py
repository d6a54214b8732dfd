"""Small problems (reference dataset sizes): kernel time vs tiles per wave and segments."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kmvp_oracle
from kernel_matrix_benchmarks_amd import _lib
for kernel in ("gaussian", "inverse-distance"):
    for n in (1000, 10000, 50000):
        y = kmvp_oracle.uniform_sphere_points(n).astype(np.float32); b = np.random.RandomState(n).randn(n, 1).astype(np.float32)
        for tiles in (0, 1, 2, 4):
            for seg in (0, 1, 4, 16):
                ctx = _lib.Context(0)
                if tiles: ctx.set_option("fast_tiles", tiles)
                if seg: ctx.set_option("segments", seg)
                ctx.set_points(y, None, _lib.KMVP_F32); ctx.set_signal(b)
                ctx.run(kernel, False); ctx.run(kernel, False)
                ks, ts = [], []
                for _ in range(10):
                    t0 = time.perf_counter(); ctx.run(kernel, False); ts.append(time.perf_counter() - t0); ks.append(ctx.last_kernel_ms)
                print(f"{kernel:17s} n={n:6d} tiles={tiles} segments={seg:2d}: kernel {np.median(ks)*1e3:7.1f} us  query {np.median(ts)*1e6:7.1f} us  {ctx.last_kernel_name}", flush=True)
                ctx.close()
