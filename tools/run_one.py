"""One configuration of the low-D kernel, a few launches (for rocprofv3).
usage: python tools/run_one.py kernel T feed n reps [segments] [dtype]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib  # noqa: E402

kernel, T, feed, n, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(float(sys.argv[4])), int(sys.argv[5])
seg = int(sys.argv[6]) if len(sys.argv) > 6 else 0
dt = sys.argv[7] if len(sys.argv) > 7 else "float32"
code, npdt = _lib.dtype_code(dt)
ctx = _lib.Context(0)
rs = np.random.RandomState(n + 3)
y = rs.rand(n, 3).astype(npdt)
b = rs.randn(n, 1).astype(npdt)
ctx.set_points(y, None, code)
ctx.set_signal(b)
ctx.set_option("feed", feed)
ctx.set_option("targets_per_lane", T)
ctx.set_option("segments", seg)
ms = []
for _ in range(reps):
    ctx.run(kernel, False)
    ms.append(ctx.last_kernel_ms)
print(kernel, "T", T, "feed", feed, "n", n, "kernel_ms", ["%.3f" % m for m in ms], "pairs/s %.3e" % (n * n / (min(ms) * 1e-3)))
