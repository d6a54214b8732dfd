"""cellmm_kernel against cell_kernel / lowd_kernel at the headline shape: error vs the C oracle and kernel ms.
usage: python tools/cmp_cellmm.py [n] [fast_tiles]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import c_oracle, kmvp_oracle
from kernel_matrix_benchmarks_amd import _lib

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 0
y, b = kmvp_oracle.uniform_cube(n, 3)
rows = np.random.RandomState(0).choice(n, size=min(n, 512), replace=False)
want = c_oracle.product(kernel="gaussian", source_points=y, source_signal=b, rows=rows)
ctx = _lib.Context(0)
ctx.set_points(y.astype(np.float32), None, _lib.KMVP_F32)
ctx.set_signal(b.astype(np.float32))
if tiles:
    ctx.set_option("fast_tiles", tiles)
res = {}
for code in (3, 4, 1, 0):
    ctx.set_option("fast_sqdists", code)
    ctx.run("gaussian", False)
    ms = []
    for _ in range(5):
        ctx.run("gaussian", False)
        ms.append(ctx.last_kernel_ms)
    a = ctx.get_result(n, 1)
    err = np.max(np.abs(a[rows] - want)) / np.max(np.abs(want))
    res[code] = a
    print(f"fast_sqdists={code} {ctx.last_kernel_name:14s} kernel {min(ms):8.3f} ms (mean {np.mean(ms):8.3f})  total {ctx.last_total_ms:8.3f} ms  "
          f"{n * n / min(ms) / 1e9:7.2f}e12 pairs/s  rel err {err:.2e}", flush=True)
print("max |cellmm - cell| / max|a| over all rows:", np.max(np.abs(res[3] - res[4])) / np.max(np.abs(res[4])))
print("max |cellmm - lowd| / max|a| over all rows:", np.max(np.abs(res[3] - res[0])) / np.max(np.abs(res[0])))
# density and a signal with a huge dynamic range
ctx.set_option("fast_sqdists", 3)
ctx.set_signal(None)
ctx.run("gaussian", False)
d = ctx.get_result(n, 1)
wd = c_oracle.product(kernel="gaussian", source_points=y, rows=rows, density_estimation=True)
print("density", ctx.last_kernel_name, "rel err", np.max(np.abs(d[rows] - wd)) / np.max(np.abs(wd)))
b2 = b * np.exp(np.random.RandomState(3).uniform(-12, 12, size=b.shape))
ctx.set_signal(b2.astype(np.float32))
ctx.run("gaussian", False)
a2 = ctx.get_result(n, 1)
w2 = c_oracle.product(kernel="gaussian", source_points=y, source_signal=b2, rows=rows)
print("wide signal", ctx.last_kernel_name, "rel err", np.max(np.abs(a2[rows] - w2)) / np.max(np.abs(w2)))
ctx.close()
