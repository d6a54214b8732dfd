"""cellmm_kernel (32x32x16 f16 MFMA) against cellmm16_kernel (16x16x32) at the headline shape (Gaussian, uniform-3D, N = M = 1e6,
float32): interleaved rounds in one process, error of both on 512 oracle rows.  usage: python tools/cellmm_shapes.py [rounds] [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd import _lib
import c_oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
rs = np.random.RandomState(n + 3)
y64 = rs.rand(n, 3); b64 = rs.randn(n, 1)
y = y64.astype(np.float32); b = b64.astype(np.float32)
rows = np.random.RandomState(0).choice(n, size=512, replace=False)
want = c_oracle.product(kernel="gaussian", source_points=y64, source_signal=b64, rows=rows)
ctx = _lib.Context(0)
ctx.set_option("fast_sqdists", 3)
ctx.set_points(y, None, _lib.KMVP_F32)
ctx.fit("gaussian")
ctx.set_signal(b)
ms = {0: [], 1: []}
res = {}
for shape in (0, 1):
    ctx.set_option("cellmm_shape", shape)
    ctx.run("gaussian", False)
    res[shape] = ctx.get_result(n, 1)
    print(f"shape {shape}: {ctx.last_kernel_name} rel err on 512 rows {np.max(np.abs(res[shape][rows] - want)) / np.max(np.abs(want)):.2e}", flush=True)
print(f"max |shape1 - shape0| / max|a| over all rows: {np.max(np.abs(res[1] - res[0])) / np.max(np.abs(res[0])):.2e}")
for _ in range(rounds):
    for shape in (0, 1):
        ctx.set_option("cellmm_shape", shape)
        for _ in range(5):
            ctx.run("gaussian", False)
            ms[shape].append(ctx.last_kernel_ms)
for shape in (0, 1):
    a = np.array(ms[shape])
    print(f"shape {shape}: kernel min {a.min():.3f} median {np.median(a):.3f} max {a.max():.3f} ms  -> {n * float(n) / (np.median(a) * 1e-3):.3e} pairs/s, "
          f"frac of 2.5 PF at 32 flop/pair {32.0 * n * n / (np.median(a) * 1e-3) / 2.5e15:.3f}")
ctx.close()
