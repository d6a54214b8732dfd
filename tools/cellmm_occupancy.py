"""What resident waves per SIMD are worth to the headline kernel: cellmm16_kernel / cellmm_kernel at TT = 4 (147-159 VGPRs:
three workgroups per CU fit) with the third workgroup allowed or crowded out by unused dynamic LDS (KMVP_DBG_LDS, read once per
process: run this script once per setting).  N = M = 1e6, Gaussian, float32.
usage: KMVP_DBG_LDS=<bytes> python tools/cellmm_occupancy.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib

n = 1000000
rs = np.random.RandomState(n + 3)
y = rs.rand(n, 3).astype(np.float32); b = rs.randn(n, 1).astype(np.float32)
ctx = _lib.Context(0)
ctx.set_option("fast_sqdists", 3)
ctx.set_points(y, None, _lib.KMVP_F32)
ctx.fit("gaussian")
ctx.set_signal(b)
for tiles in (4, 8):
    for shape in (0, 1):
        ctx.set_option("fast_tiles", tiles)
        ctx.set_option("cellmm_shape", shape)
        ms = []
        for _ in range(12):
            ctx.run("gaussian", False)
            ms.append(ctx.last_kernel_ms)
        print(f"KMVP_DBG_LDS={os.environ.get('KMVP_DBG_LDS', '0'):>6s}  TT={tiles} shape={shape} {ctx.last_kernel_name:16s} median {np.median(ms[2:]):7.3f} ms", flush=True)
ctx.close()
