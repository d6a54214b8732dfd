"""Query latency at the reference's own dataset sizes (product-sphere / product-cube, n = 1e3 .. 1e4, D = 3, E = 1)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kmvp_oracle
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct
for kernel, gen in (("inverse-distance", "sphere"), ("gaussian", "sphere")):
    for n in (1000, 2000, 5000, 10000):
        y = kmvp_oracle.uniform_sphere_points(n); b = np.random.RandomState(n).randn(n, 1)
        for prec in ("float32", "float64"):
            algo = MI355XProduct(kernel=kernel, dimension=3, precision=prec)
            algo.prepare_data(source_points=y, target_points=y, same_points=True)
            algo.prepare_query(source_signal=b)
            algo.query(); algo.query()
            ts = []
            for _ in range(20):
                t0 = time.perf_counter(); algo.query(); ts.append(time.perf_counter() - t0)
            a = algo.get_result()
            want = kmvp_oracle.product(kernel=kernel, source_points=y, source_signal=b)
            err = np.max(np.abs(a - want)) / np.max(np.abs(want))
            print(f"{kernel:17s} n={n:6d} {prec}: query {np.median(ts)*1e6:8.1f} us (kernel {algo.device_kernel_ms*1e3:7.1f} us, {algo.device_kernel}) rel err {err:.1e}", flush=True)
            algo.done()
