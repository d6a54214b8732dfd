"""GPU exploration: correctness spot-check + tuning sweep of the low-D kernel.
Run on the GPU box:  python tools/explore.py [n]   (writes gpurun_out/explore.log)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd import _lib  # noqa: E402
import c_oracle  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
    ctx = _lib.Context(0)
    rs = np.random.RandomState(n + 3)
    y64 = rs.rand(n, 3)
    b64 = rs.randn(n, 1)
    y = y64.astype(np.float32)
    b = b64.astype(np.float32)
    rows = np.random.RandomState(1).choice(n, size=512, replace=False)
    t0 = time.time()
    ref = c_oracle.product(kernel="gaussian", source_points=y64, source_signal=b64, rows=rows)
    print(f"oracle rows: {time.time()-t0:.2f}s  threads={c_oracle.threads()}", flush=True)
    ctx.set_points(y, None, _lib.KMVP_F32)
    ctx.set_signal(b)
    for kernel in ("gaussian", "inverse-distance", "absolute-exponential"):
        if kernel != "gaussian":
            ref_k = c_oracle.product(kernel=kernel, source_points=y64, source_signal=b64, rows=rows)
        else:
            ref_k = ref
        for feed in (0, 1):
            for T in (1, 2, 4, 8):
                for seg in (0,):
                    ctx.set_option("feed", feed)
                    ctx.set_option("targets_per_lane", T)
                    ctx.set_option("segments", seg)
                    try:
                        ctx.run(kernel, False)
                    except _lib.KmvpError as e:
                        print(kernel, feed, T, "ERR", e, flush=True)
                        continue
                    best = 1e9
                    for _ in range(3):
                        ctx.run(kernel, False)
                        best = min(best, ctx.last_kernel_ms)
                    out = ctx.get_result(n, 1)
                    err = np.max(np.abs(out[rows] - ref_k)) / np.max(np.abs(ref_k))
                    print(f"{kernel:22s} feed={feed} T={T} seg=auto kernel_ms={best:9.3f} total_ms={ctx.last_total_ms:9.3f} "
                          f"pairs/s={n*n/(best*1e-3):.3e} relerr={err:.2e}", flush=True)
    # segments sweep on the best-known config
    ctx.set_option("feed", 0)
    ctx.set_option("targets_per_lane", 4)
    for seg in (1, 8, 16, 32, 64, 128, 256):
        ctx.set_option("segments", seg)
        ctx.run("gaussian", False)
        best = 1e9
        for _ in range(3):
            ctx.run("gaussian", False)
            best = min(best, ctx.last_kernel_ms)
        print(f"gaussian feed=0 T=4 seg={seg:4d} kernel_ms={best:9.3f} total_ms={ctx.last_total_ms:9.3f} pairs/s={n*n/(best*1e-3):.3e}", flush=True)
    for chunk in (64, 256, 1024, 4096):
        ctx.set_option("segments", 0)
        ctx.set_option("chunk", chunk)
        ctx.run("gaussian", False)
        best = 1e9
        for _ in range(3):
            ctx.run("gaussian", False)
            best = min(best, ctx.last_kernel_ms)
        out = ctx.get_result(n, 1)
        err = np.max(np.abs(out[rows] - ref)) / np.max(np.abs(ref))
        print(f"gaussian chunk={chunk:5d} kernel_ms={best:9.3f} relerr={err:.2e}", flush=True)


if __name__ == "__main__":
    main()
