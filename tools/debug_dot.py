"""exp(<x,y>) native kernel on extreme logits: which rows go wrong, and what their logits look like."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kmvp_oracle
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct

rs = np.random.RandomState(2024)
f32 = lambda a: a.astype(np.float32).astype(np.float64)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
M, N = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (500, 64)
y, x, b = f32(rs.randn(M, 4) * scale), f32(rs.randn(N, 4) * scale), f32(rs.randn(M, 2))
L = (x @ y.T) * 1.4426950408889634
for seg in (0, 1):
    algo = MI355XProduct(kernel="exp-dot", dimension=4, normalize_rows=True, precision="float32", segments=seg)
    algo.prepare_data(source_points=y, target_points=x, same_points=False)
    algo.prepare_query(source_signal=b)
    algo.query()
    got = algo.get_result()
    algo.done()
    want = kmvp_oracle.exp_dot_product(source_points=y, target_points=x, source_signal=b, normalize_rows=True)
    bad = np.where(~np.isfinite(got).all(axis=1))[0]
    print(f"segments={seg}: non-finite rows {bad.tolist()}")
    for i in bad[:6]:
        o = np.argsort(-L[i])
        print(f"  row {i}: top logits (log2) {L[i][o[:4]].round(1).tolist()} at sources {o[:4].tolist()}, min {L[i].min():.0f}, got {got[i]}")
    ok = np.isfinite(got).all(axis=1)
    print("  finite rows max err", np.max(np.abs(got[ok] - want[ok])))
