"""What the per-target running shift costs the bf16 Gaussian with targets != sources (N = M = 65536, D = E = 64, normalised):
the default (K_GAUSSIAN_SHIFTED) against the plain kernel forced by mfma_variant = 0.  Measured (round 3): 1.170 against 1.048 ms.
usage: python tools/bf16_shift_cost.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct
n, D, E = 65536, 64, 64
rs = np.random.RandomState(1)
y = rs.rand(n, D) / np.sqrt(D); x = rs.rand(n, D) / np.sqrt(D); b = rs.randn(n, E)
for variant in (-1, 0):
    algo = MI355XProduct(kernel="gaussian", dimension=D, normalize_rows=True, precision="bfloat16")
    algo.prepare_data(source_points=y, target_points=x, same_points=False)
    algo.set_query_arguments(mfma_variant=variant)
    algo.fit(); algo.prepare_query(source_signal=b)
    for _ in range(50): algo.query()
    ms = []
    for _ in range(20):
        algo.query(); ms.append(algo.device_kernel_ms)
    print("mfma_variant", variant, algo.device_kernel, round(float(np.mean(ms)), 4), "ms", algo.get_additional().get("dispatch_note", "")[:70])
    algo.done()
