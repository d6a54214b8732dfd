"""query() wall time against problem size: what the reference's runner books as query_time (runner.py:83-96) for the
datasets of its list (product-cube-D3-E1, N = M = 1e3 ... 1e6), per kernel.  Shows where a step is launch- and
synchronisation-bound rather than kernel-bound.

    python tools/latency_sweep.py [--sizes 1000,10000,100000,1000000] [--precision float32]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--sizes", default="1000,10000,100000,1000000")
    p.add_argument("--precision", default="float32")
    p.add_argument("--D", type=int, default=3)
    p.add_argument("--E", type=int, default=1)
    a = p.parse_args()
    from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct

    print(f"{'kernel':22s} {'N = M':>9s} {'device kernel':>22s} {'kernel ms':>10s} {'device step ms':>14s} {'query() wall ms':>16s} "
          f"{'get_result ms':>13s}")
    for n in [int(v) for v in a.sizes.split(",")]:
        rs = np.random.RandomState(n + a.D)
        y = rs.rand(n, a.D)
        b = rs.randn(n, a.E)
        for kernel in ("gaussian", "absolute-exponential", "inverse-distance"):
            if kernel != "gaussian" and n > 300000:
                reps = 2
            else:
                reps = 20 if n <= 100000 else 5
            algo = MI355XProduct(kernel=kernel, dimension=a.D, precision=a.precision)
            algo.prepare_data(source_points=y, target_points=y, same_points=True)
            algo.fit()
            algo.prepare_query(source_signal=b)
            for _ in range(3):
                algo.query()
            t0 = time.perf_counter()
            for _ in range(reps):
                algo.query()
            wall = (time.perf_counter() - t0) / reps * 1e3
            t0 = time.perf_counter()
            algo.get_result()
            tg = (time.perf_counter() - t0) * 1e3
            print(f"{kernel:22s} {n:9d} {algo.device_kernel:>22s} {algo.device_kernel_ms:10.4f} {algo.device_total_ms:14.4f} "
                  f"{wall:16.4f} {tg:13.4f}", flush=True)
            algo.done()


if __name__ == "__main__":
    main()
