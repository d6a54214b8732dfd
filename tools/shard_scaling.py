"""Single-GPU rehearsal of the strong-scaling bench: rank 0's share of the headline product for
world = 1, 2, 4, 8 (all N targets x M/world sources, no all-reduce -- there is one GPU here).
Prints wall time per query, device time of the whole step and of the pair loop alone, and the
efficiency those imply BEFORE the RCCL all-reduce (8 MB fp64 at N = 1e6)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib, sharding

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
kernel = sys.argv[2] if len(sys.argv) > 2 else "gaussian"
rs = np.random.RandomState(n + 3)
y = rs.rand(n, 3).astype(np.float32); b = rs.randn(n, 1).astype(np.float32)
# argv[3] = "caller": shard in the caller's order even where the plugin would shard cell by cell (Gaussian)
order = sharding.spatial_order(y) if kernel == "gaussian" and (len(sys.argv) <= 3 or sys.argv[3] != "caller") else None
ys, bs = (y, b) if order is None else (y[order], b[order])
print("sources sharded", "in the caller's order" if order is None else "cell by cell (sharding.spatial_order)")
base = None
for world in (1, 2, 4, 8):
    lo, hi = 0, (n + world - 1) // world
    ctx = _lib.Context(0)
    ctx.set_option("same_points_global", 1)
    ctx.set_option("partial_shard", 1)  # a shard without a communicator, on purpose
    ctx.set_points(np.ascontiguousarray(ys[lo:hi]), y, _lib.KMVP_F32, j_offset=lo, M_total=n)
    ctx.set_signal(np.ascontiguousarray(bs[lo:hi]))
    ctx.run(kernel, False); ctx.run(kernel, False)
    wall, kms, tms = [], [], []
    for _ in range(10):
        t0 = time.perf_counter(); ctx.run(kernel, False); wall.append((time.perf_counter() - t0) * 1e3)
        kms.append(ctx.last_kernel_ms); tms.append(ctx.last_total_ms)
    w, k, t = np.mean(wall), np.mean(kms), np.mean(tms)
    if base is None: base = w
    print(f"world {world}: M_shard {hi-lo:8d} {ctx.last_kernel_name:13s} wall {w:8.3f} ms  device step {t:8.3f} ms  pair loop {k:8.3f} ms  "
          f"efficiency before all-reduce {base / (world * w):.3f}", flush=True)
    ctx.close()
