"""Config 5 by mixed-precision refinement at several inner tolerances: seconds, inner iterations, refinement steps, float64 residual."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, c_oracle, kmvp_oracle
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct, MI355XSolver
n = 100_000
y, b = kmvp_oracle.uniform_cube(n, 3)
prod = MI355XProduct(kernel="gaussian", dimension=3, precision=np.float64)
prod.prepare_data(source_points=y, target_points=y, same_points=True); prod.fit(); prod.prepare_query(source_signal=b); prod.query()
a = prod.get_result(); prod.done()
rows = np.random.RandomState(3).choice(n, size=256, replace=False)
for kw in (dict(), dict(refine="float32", inner_rtol=1e-2), dict(refine="float32", inner_rtol=1e-3), dict(refine="float32", inner_rtol=1e-4)):
    sol = MI355XSolver(kernel="gaussian", dimension=3, precision=np.float64, rtol=1e-6, maxit=5000, **kw)
    sol.prepare_data(source_points=y); sol.fit(); sol.prepare_query(target_signal=a)
    sol.query()
    t0 = time.perf_counter(); sol.query(); dt = time.perf_counter() - t0
    x = sol.get_result(); info = sol.get_additional()
    Kb = c_oracle.product(kernel="gaussian", source_points=y, source_signal=x, rows=rows)
    res = np.linalg.norm(Kb - a[rows]) / np.linalg.norm(a[rows])
    print(kw, f"{dt*1e3:.1f} ms", info, f"oracle rows residual {res:.2e}", flush=True)
    sol.done()
