"""fastmm_kernel, D = 1, E = 17, x != y: column errors with the online shift and (forced) without."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kmvp_oracle
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct

D, E, norm = 1, 17, False
rng = np.random.RandomState(100 * D + E)
n, m = 1237, 2051
y = rng.rand(m, D) / np.sqrt(max(D, 3) / 3.0)
x = rng.rand(n, D) / np.sqrt(max(D, 3) / 3.0)
b = rng.randn(m, E) * 10.0 ** rng.randint(-6, 7, size=E)
want = kmvp_oracle.product(kernel="gaussian", source_points=y, target_points=x, source_signal=b, normalize_rows=norm)
for forced_offline in (0, 1):
    for tiles in (1, 2):
        algo = MI355XProduct(kernel="gaussian", dimension=D, normalize_rows=norm, precision="float32", fast_sqdists=True, fast_tiles=tiles)
        algo.prepare_data(source_points=y, target_points=x)
        if forced_offline:
            algo._ctx.set_option("same_points_global", 1)
        algo.prepare_query(source_signal=b)
        algo.query()
        got = algo.get_result()
        note = algo.get_additional()["dispatch_note"]
        algo.done()
        col_err = np.abs(got - want).max(axis=0) / np.abs(want).max(axis=0)
        worst = int(np.argmax(col_err))
        i = int(np.argmax(np.abs(got[:, worst] - want[:, worst])))
        print(f"offline={forced_offline} tiles={tiles} note={note!r}\n  col_err max {col_err.max():.2e} (column {worst}, row {i}: got {got[i, worst]:.9e} want {want[i, worst]:.9e}); sorted {np.sort(col_err)[-4:]}")
