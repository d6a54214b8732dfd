"""Beyond the BASELINE sizes: Gaussian product of 1e7 targets x 2e6 sources (targets != sources, float32) through the plugin,
256 rows against the C oracle -- index widths, buffer sizes and the cell lists at 2e13 pairs.  Measured (round 3): cellmm16_kernel
465 ms, 4.3e13 pairs/s, relative error 1.3e-7, 2.2 GB on the device.  usage: python tools/big_gaussian_check.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct
import c_oracle
rs = np.random.RandomState(7)
N, M = 10_000_000, 2_000_000
y = rs.rand(M, 3); x = rs.rand(N, 3); b = rs.randn(M, 1)
algo = MI355XProduct(kernel="gaussian", dimension=3, precision="float32")
algo.prepare_data(source_points=y, target_points=x, same_points=False)
algo.fit(); algo.prepare_query(source_signal=b)
algo.query(); t0 = time.time(); algo.query(); dt = time.time() - t0
got = algo.get_result()
rows = np.sort(rs.choice(N, size=256, replace=False))
want = c_oracle.product(kernel="gaussian", source_points=y, target_points=x, source_signal=b, rows=rows)
print(algo.device_kernel, algo.device_kernel_ms, "ms; wall", dt, "rel err", np.abs(got[rows] - want).max() / np.abs(want).max(), "finite", np.isfinite(got).all(),
      "pairs/s", N * M / (algo.device_kernel_ms * 1e-3), "device MB", algo.get_additional().get("device_bytes", 0) / 1e6)
algo.done()
