import os, sys
import numpy as np
sys.argv = sys.argv[:1]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "debug_cell.py")).read().split("rs = np.random.RandomState(0)")[0])
rs = np.random.RandomState(5)
h = np.sqrt(2 * 0.006 / 3)
x = rs.rand(60, 3); y = x[:1].copy()
got, name = run(y, np.ones((1, 1)), x); want = ref(y, np.ones((1, 1)), x)
lo = np.minimum(x.min(0), y.min(0))
cells = np.floor((x.astype(np.float32) - lo.astype(np.float32)) / np.float32(h)).astype(int)
order = np.lexsort((cells[:, 0], cells[:, 1], cells[:, 2]))
for rank, i in enumerate(order):
    print(rank, i, cells[i], f"got {got[i]:.6f} want {want[i]:.6f} ratio {got[i]/want[i]:.6f}")
