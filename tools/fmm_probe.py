"""fastmm_kernel probe: accuracy against the float64 oracle on a row sample and device time, for a few shapes and
column counts, beside the per-column cell form and the column-blocked difference form.
usage: python tools/fmm_probe.py [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct  # noqa: E402
from oracle import c_oracle, kmvp_oracle  # noqa: E402  (checker only)


def rel_err(got, want):
    return float(np.abs(got - want).max() / np.abs(want).max())


KERNEL = os.environ.get("FMM_PROBE_KERNEL", "gaussian")
SPREAD = float(os.environ.get("FMM_PROBE_SPREAD", "1"))  # > 1: clouds outside fast_kernel's radius rule


def run(n, D, E, norm, form, tiles=0, segments=0):
    y, b = kmvp_oracle.uniform_cube(n, D, E=E)
    y = y * SPREAD
    algo = MI355XProduct(kernel=KERNEL, dimension=D, normalize_rows=norm, precision="float32", fast_sqdists=form,
                         fast_tiles=tiles, segments=segments)
    try:
        algo.prepare_data(source_points=y, target_points=y, same_points=True)
        algo.fit()
        algo.prepare_query(source_signal=b)
        algo.query()
        best = 1e30
        for _ in range(3):
            algo.query()
            best = min(best, algo.device_kernel_ms)
        got = algo.get_result()
        kname = algo.device_kernel
    finally:
        algo.done()
    rows = np.random.RandomState(8).choice(n, size=min(n, 128), replace=False)
    want = c_oracle.product(kernel=KERNEL, source_points=y, source_signal=b, rows=rows, normalize_rows=norm)
    return kname, best, rel_err(got[rows], want), float(np.abs((got[rows] - want) / np.abs(want).max(axis=0)).max())


if __name__ == "__main__":
    sizes = [int(float(v)) for v in sys.argv[1:]] or [100_000]
    for n in sizes:
        shapes = os.environ.get("FMM_PROBE_SHAPES")
        shapes = [tuple(int(v) for v in sh.split(",")) for sh in shapes.split(";")] if shapes else None
        for D, E, norm in shapes or ((3, 16, True), (3, 2, False), (3, 8, False), (3, 15, True), (3, 31, True), (3, 40, True), (2, 4, True), (5, 16, True), (8, 16, False)):
            forms = ((None, 0), (True, 2)) if os.environ.get("FMM_PROBE_SHORT") == "1" else ((True, 1), (True, 2), (True, 4)) if os.environ.get("FMM_PROBE_SHORT") == "2" else ((None, 0), ("centred", 1), ("centred", 2), (False, 0)) if os.environ.get("FMM_PROBE_SHORT") == "3" else (
                (None, 0), (True, 1), (True, 2), (True, 4), ("cells", 0), (False, 0))
            if os.environ.get("FMM_PROBE_SEGMENTS"):
                for seg in (int(v) for v in os.environ["FMM_PROBE_SEGMENTS"].split(",")):
                    k, ms, err, colerr = run(n, D, E, norm, True, 2, seg)
                    print(f"n={n} D={D} E={E} norm={int(norm)} segments={seg:3d} -> {k:14} {ms:9.3f} ms  err {err:.2e}", flush=True)
                continue
            for form, tiles in forms:
                if n > 200_000 and form is False:
                    continue
                try:
                    t0 = time.time()
                    k, ms, err, colerr = run(n, D, E, norm, form, tiles)
                    print(f"n={n} D={D} E={E} norm={int(norm)} form={form!s:6} tiles={tiles} -> {k:14} {ms:9.3f} ms  "
                          f"err {err:.2e}  per-column err {colerr:.2e}  ({time.time() - t0:.1f} s)", flush=True)
                except Exception as exc:  # noqa: BLE001
                    print(f"n={n} D={D} E={E} norm={int(norm)} form={form} tiles={tiles}: {type(exc).__name__}: {exc}", flush=True)
