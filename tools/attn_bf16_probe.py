"""Attention shapes at N = M = 65536, D = E = 64, bfloat16: exp(-r) (BASELINE config 3), Gaussian and exp<x, y> (softmax
attention), kernel time of each.  usage: python tools/attn_bf16_probe.py [n=65536]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct  # noqa: E402
import kmvp_oracle  # noqa: E402  (checker only)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D = E = 64
rs = np.random.RandomState(n + D)
y = rs.rand(n, D) / np.sqrt(D)
b = rs.randn(n, E)
rows = np.sort(rs.choice(n, size=128, replace=False))
# (the last line: exp<x, y> on the SAME narrow cloud as the first two -- the launches are power-limited and the operand bits
# of nearly equal weights toggle less than those of a real softmax)
for kernel, pts in (("absolute-exponential", y), ("gaussian", y), ("exp-dot", (rs.randn(n, D) * 0.35)), ("exp-dot", rs.randn(n, D)),
                    ("exp-dot", y)):
    algo = MI355XProduct(kernel=kernel, dimension=D, normalize_rows=True, precision="bfloat16")
    try:
        algo.prepare_data(source_points=pts, target_points=pts, same_points=True)
        algo.fit()
        algo.prepare_query(source_signal=b)
        for _ in range(50):
            algo.query()
        ms = []
        for _ in range(20):
            algo.query()
            ms.append(algo.device_kernel_ms)
        got = algo.get_result()[rows]
        if kernel == "exp-dot":
            want = kmvp_oracle.exp_dot_product(source_points=pts, target_points=pts[rows], source_signal=b, normalize_rows=True)
        else:
            want = kmvp_oracle.product(kernel=kernel, source_points=pts, source_signal=b, normalize_rows=True, rows=rows)
        err = float(np.abs(got - want).max() / np.abs(want).max())
        flop = 2.0 * (D + E + 1) * n * n
        print(f"{kernel:22s} {algo.device_kernel:18s} {np.mean(ms):7.3f} ms  {flop / (np.mean(ms) * 1e-3) / 1e12:7.1f} TFLOP/s  rel err {err:.2e}  "
              f"{algo.get_additional().get('dispatch_note', '')[:160]}", flush=True)
    except NotImplementedError as e:
        print(f"{kernel:22s} NotImplementedError: {str(e)[:200]}", flush=True)
    finally:
        algo.done()
