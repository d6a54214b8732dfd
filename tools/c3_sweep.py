"""Config 3 (exp(-r) attention, N = M = 65536, D = 64, E = 64, bf16) against the number of source segments and the
pipelined / plain kernel: kernel ms (min, mean of 10) and device ms of the whole step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct
n, D, E = 65536, 64, 64
rs = np.random.RandomState(n + D)
y = rs.rand(n, D) / np.sqrt(D); b = rs.randn(n, E)
for seg in (0, 2, 4, 8, 16, 32):
    for T in (0, 2):
        algo = MI355XProduct(kernel="absolute-exponential", dimension=D, normalize_rows=True, precision="bfloat16", segments=seg, targets_per_lane=T)
        algo.prepare_data(source_points=y, target_points=y, same_points=True); algo.prepare_query(source_signal=b)
        algo.query(); ms = []
        for _ in range(10):
            algo.query(); ms.append(algo.device_kernel_ms)
        print(f"segments {seg:2d} T {T}: {algo.device_kernel} kernel min {min(ms):.3f} mean {np.mean(ms):.3f} ms  total {algo.device_total_ms:.3f}", flush=True)
        algo.done()
