"""Config 3 (exp(-r) attention, N = M = 65536, D = 64, E = 64, bf16) over the variants of mfma_pipe_kernel
(option mfma_variant), interleaved rounds in ONE process (guide rule 24); error of every variant on 64 oracle rows.
usage: python tools/c3_variants.py [rounds] [variants, comma separated] [kernel]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd import _lib
import c_oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
# variant codes: 0, 1, 4, 5 = mfma_pipe_kernel VAR; 11 / 12 = mfma_kernel with one / two target tiles per wave (no software pipelining)
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1,4,5,11,12").split(",")]
kernel = sys.argv[3] if len(sys.argv) > 3 else "absolute-exponential"
zero = len(sys.argv) > 4 and sys.argv[4] == "zero"  # all-zero operands: the chip holds ~2.4 GHz, times compare CYCLES
n, D, E = 65536, 64, 64
rs = np.random.RandomState(n + D)
y64 = rs.rand(n, D) / np.sqrt(D); b64 = rs.randn(n, E)
if zero:
    y64 = np.zeros_like(y64); b64 = np.zeros_like(b64)
y = y64.astype(np.float32); b = b64.astype(np.float32)
rows = np.random.RandomState(0).choice(n, size=64, replace=False)
want = c_oracle.product(kernel=kernel, source_points=y64, source_signal=b64, rows=rows, normalize_rows=True)
ctx = _lib.Context(0)
ctx.set_points(y, None, _lib.KMVP_BF16)
ctx.set_signal(b)
ms = {v: [] for v in variants}
err = {}
def select(v):
    ctx.set_option("targets_per_lane", v - 10 if v >= 10 else 0)
    ctx.set_option("mfma_variant", v if v < 10 else 0)


for v in variants:  # warm-up + error
    select(v)
    ctx.run(kernel, True)
    got = ctx.get_result(n, E)
    err[v] = float(np.max(np.abs(got[rows] - want)) / max(np.max(np.abs(want)), 1e-300))
for _ in range(rounds):
    for v in variants:
        select(v)
        ctx.run(kernel, True)  # (re-packs when the tile count changed)
        for _ in range(3):
            ctx.run(kernel, True)
            ms[v].append(ctx.last_kernel_ms)
for v in variants:
    a = np.array(ms[v])
    print(f"variant {v}: min {a.min():.4f} median {np.median(a):.4f} max {a.max():.4f} ms  "
          f"frac(258 flop/pair, 2.5 PF) at median {258.0 * n * n / (np.median(a) * 1e-3) / 2.5e15:.3f}  rel err {err[v]:.2e}", flush=True)
ctx.close()
