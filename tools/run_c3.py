"""Config 3 (exp(-r) attention, N = M = 65536, D = 64, E = 64, bf16) a few launches, for
rocprofv3 and tuning.  usage: python tools/run_c3.py [reps] [targets_per_lane] [segments] [kernel] [normalize]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
T = int(sys.argv[2]) if len(sys.argv) > 2 else 0
seg = int(sys.argv[3]) if len(sys.argv) > 3 else 8
kernel = sys.argv[4] if len(sys.argv) > 4 else "absolute-exponential"
norm = (sys.argv[5] != "0") if len(sys.argv) > 5 else True
n, D, E = 65536, 64, 64
rs = np.random.RandomState(n + D)
y = (rs.rand(n, D) / np.sqrt(D)).astype(np.float32); b = rs.randn(n, E).astype(np.float32)
ctx = _lib.Context(0)
ctx.set_option("segments", seg)
if T: ctx.set_option("targets_per_lane", T)
ctx.set_points(y, None, _lib.KMVP_BF16)
ctx.set_signal(b)
ms = []
for _ in range(reps):
    ctx.run(kernel, norm); ms.append(ctx.last_kernel_ms)
k = min(ms)
print(f"{kernel} norm={norm} T={T} segments={seg} {ctx.last_kernel_name}: kernel_ms {['%.3f' % m for m in ms]} pairs/s {n*n/(k*1e-3):.3e} "
      f"MFMA {2.0*n*n*(80+64)/(k*1e-3)/1e15:.3f} PFLOP/s", flush=True)
ctx.close()
