"""Debugging aid: bf16 exp(-r) attention at small shapes under every mfma_variant, twice each (run-to-run equality, error against the oracle)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kmvp_oracle
from kernel_matrix_benchmarks_amd import _lib
for (n, D, E) in ((300, 32, 5), (300, 32, 33), (1000, 64, 64), (257, 16, 3)):
    yd = np.random.RandomState(5).rand(n, D) / np.sqrt(D)
    bd = np.random.RandomState(6).randn(n, E)
    want = kmvp_oracle.product(kernel="absolute-exponential", source_points=yd, source_signal=bd, normalize_rows=True)
    for v in (0, 1, 4, 5):
        for rep in range(2):
            ctx = _lib.Context(0)
            ctx.set_option("mfma_variant", v)
            ctx.set_points(yd.astype(np.float32), None, _lib.KMVP_BF16)
            ctx.set_signal(bd.astype(np.float32))
            ctx.run("absolute-exponential", True)
            got = ctx.get_result(n, E)
            ctx.close()
            bad = np.where(~np.isfinite(got).all(axis=1))[0]
            ok = np.isfinite(got).all(axis=1)
            print(f"n={n} D={D} E={E} variant {v} rep {rep}: bad rows {bad[:12].tolist()} ({len(bad)}), err on the rest {np.max(np.abs(got[ok]-want[ok]))/np.max(np.abs(want)):.2e}")
