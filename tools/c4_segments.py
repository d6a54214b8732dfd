"""Config 4, one of 8 shards (1e7 targets x 1.25e6 sources, 1/r, float32, cfast_kernel) against the number of source
segments: pair-loop ms, whole-step ms (incl. the reduction of the fp64 partial sums) and the partial-sum traffic
segments x N x 8 B written + read back.  usage: python tools/c4_segments.py [segments, comma separated]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib

segs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,8,16,24,32,48,64").split(",")]
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
whole = len(sys.argv) > 3 and sys.argv[3] == "whole"  # all sources on this GPU (the C2 shape with 1/r) instead of one of 8 shards
rs = np.random.RandomState(n + 3)
y = rs.rand(n, 3).astype(np.float32)
b = rs.randn(n, 1).astype(np.float32)
lo, hi = (0, n) if whole else (n * 3 // 8, n * 4 // 8)
ctx = _lib.Context(0)
ctx.set_option("same_points_global", 1)
ctx.set_option("partial_shard", 1)
ctx.set_points(np.ascontiguousarray(y[lo:hi]), y, _lib.KMVP_F32, j_offset=lo, M_total=n)
ctx.set_signal(np.ascontiguousarray(b[lo:hi]))
ref = None
for s in segs:
    ctx.set_option("segments", s)
    ctx.run("inverse-distance", False)
    ms, tot = [], []
    for _ in range(2):
        ctx.run("inverse-distance", False)
        ms.append(ctx.last_kernel_ms); tot.append(ctx.last_total_ms)
    got = ctx.get_result(n, 1)
    if ref is None:
        ref = got
    print(f"segments={s:3d} {ctx.last_kernel_name}: kernel {min(ms):8.2f} ms  step {min(tot):8.2f} ms  device {ctx.device_bytes/1e9:.2f} GB  "
          f"max |diff| vs first {np.max(np.abs(got - ref)) / np.max(np.abs(ref)):.1e}", flush=True)
ctx.close()
