"""bf16 MFMA path: accuracy on small shapes vs the fp64 oracle, timing at BASELINE config 3."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd import _lib
import kmvp_oracle, c_oracle

def rel(a, b):
    return np.max(np.sqrt(np.sum((a-b)**2, -1))) / np.max(np.sqrt(np.sum(b**2, -1)))

ctx = _lib.Context(0)
for (N, M, D, E) in ((96, 160, 64, 64), (200, 333, 16, 5), (64, 64, 128, 16), (257, 193, 3, 1), (1000, 1000, 64, 64)):
    rs = np.random.RandomState(N + D)
    y = rs.rand(M, D) / np.sqrt(D); x = rs.rand(N, D) / np.sqrt(D); b = rs.randn(M, E)
    for kernel in ("gaussian", "absolute-exponential", "inverse-distance"):
        for nr in (False, True):
            want = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b, normalize_rows=nr)
            ctx.set_points(y.astype(np.float32), x.astype(np.float32), _lib.KMVP_BF16)
            ctx.set_signal(b.astype(np.float32))
            ctx.run(kernel, nr)
            got = ctx.get_result(N, E)
            print(f"N={N} M={M} D={D} E={E} {kernel:22s} nr={nr!s:5s} rel={rel(got, want):.2e}", flush=True)

n, D, E = 65536, 64, 64
rs = np.random.RandomState(n + D)
y = (rs.rand(n, D) / np.sqrt(D)); b = rs.randn(n, E)
rows = np.random.RandomState(0).choice(n, 256, replace=False)
for kernel in ("absolute-exponential", "gaussian", "inverse-distance"):
    want = c_oracle.product(kernel=kernel, source_points=y, source_signal=b, normalize_rows=True, rows=rows)
    ctx.set_points(y.astype(np.float32), None, _lib.KMVP_BF16)
    ctx.set_signal(b.astype(np.float32))
    for seg in (0, 1, 2, 8):
        ctx.set_option("segments", seg)
        ctx.run(kernel, True)
        best = 1e9
        for _ in range(5):
            ctx.run(kernel, True); best = min(best, ctx.last_kernel_ms)
        got = ctx.get_result(n, E)
        print(f"C3 {kernel:22s} seg={seg} kernel_ms={best:.3f} total_ms={ctx.last_total_ms:.3f} pairs/s={n*n/(best*1e-3):.3e} rel={rel(got[rows], want):.2e}", flush=True)
# f32 generic path for comparison (slow but exact)
ctx.set_option("segments", 0)
ctx.set_points(y.astype(np.float32), None, _lib.KMVP_F32); ctx.set_signal(b.astype(np.float32))
t0 = time.time(); ctx.run("absolute-exponential", True); print("f32 generic C3: kernel_ms", ctx.last_kernel_ms)
