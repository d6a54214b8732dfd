"""Wall time of every plugin phase at a BASELINE shape (prepare_data = H2D + bounding box, fit = whatever the points alone
determine, prepare_query = signal H2D, first query = packing + product, later queries).  usage: phase_times.py [n] [E]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct  # noqa: E402
import kmvp_oracle  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 1
y, b = kmvp_oracle.uniform_cube(n, 3, E=E)
for rep in range(2):
    t = [time.perf_counter()]
    algo = MI355XProduct(kernel="gaussian", dimension=3, precision="float32")
    algo.prepare_data(source_points=y, target_points=y, same_points=True); t.append(time.perf_counter())
    algo.fit(); t.append(time.perf_counter())
    algo.prepare_query(source_signal=b); t.append(time.perf_counter())
    algo.query(); t.append(time.perf_counter())
    algo.query(); t.append(time.perf_counter())
    a = algo.get_result(); t.append(time.perf_counter())
    k = algo.device_kernel
    algo.done()
    names = ["prepare_data", "fit", "prepare_query", "query#1", "query#2", "get_result"]
    print(f"rep {rep} n={n} E={E} {k}: " + "  ".join(f"{nm} {1e3 * (t[i + 1] - t[i]):.1f} ms" for i, nm in enumerate(names)), flush=True)
