"""Error of config 2's forms (cells on the matrix cores / on the VALU / expanded) on 4096 oracle rows, next to what rounding the inputs to float32 alone costs."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, c_oracle, kmvp_oracle
from kernel_matrix_benchmarks_amd import _lib
n = 1_000_000
y, b = kmvp_oracle.uniform_cube(n, 3)
rows = np.random.RandomState(0).choice(n, size=4096, replace=False)
y32 = y.astype(np.float32); b32 = b.astype(np.float32)
want = c_oracle.product(kernel="gaussian", source_points=y, source_signal=b, rows=rows)
want32 = c_oracle.product(kernel="gaussian", source_points=y32.astype(np.float64), source_signal=b32.astype(np.float64), rows=rows)
print("input rounding alone (oracle on float32-rounded inputs vs original): %.2e" % (np.max(np.abs(want32 - want)) / np.max(np.abs(want))))
ctx = _lib.Context(0); ctx.set_points(y32, None, _lib.KMVP_F32); ctx.set_signal(b32)
for code in (3, 4, 1):
    ctx.set_option("fast_sqdists", code); ctx.run("gaussian", False); a = ctx.get_result(n, 1)
    e = np.abs(a[rows] - want) / np.max(np.abs(want)); e32 = np.abs(a[rows] - want32) / np.max(np.abs(want))
    print(f"{ctx.last_kernel_name:14s} vs truth: max {e.max():.2e} rms {np.sqrt(np.mean(e**2)):.2e} | vs truth on the rounded inputs: max {e32.max():.2e} rms {np.sqrt(np.mean(e32**2)):.2e}")
ctx.close()
