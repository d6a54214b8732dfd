"""Config 3's kernel on random against all-zero points / signals: how much of its time is the chip's power limit (the same instruction stream on zero operands runs at the full clock)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct
n, D, E = 65536, 64, 64
rs = np.random.RandomState(n + D)
for label, y, b in (("random", rs.rand(n, D) / np.sqrt(D), rs.randn(n, E)), ("zero points", np.zeros((n, D)), rs.randn(n, E)),
                    ("zero signal", rs.rand(n, D) / np.sqrt(D), np.zeros((n, E))), ("all zero", np.zeros((n, D)), np.zeros((n, E)))):
    algo = MI355XProduct(kernel="absolute-exponential", dimension=D, normalize_rows=True, precision="bfloat16")
    algo.prepare_data(source_points=y, target_points=y, same_points=True); algo.prepare_query(source_signal=b)
    for _ in range(30): algo.query()
    ms = []
    for _ in range(30):
        algo.query(); ms.append(algo.device_kernel_ms)
    print(f"{label:12s}: {algo.device_kernel} kernel min {min(ms):.3f} mean {np.mean(ms):.3f} ms", flush=True)
    algo.done()
