"""Solver latency at the reference's dataset sizes (solver-sphere / solver-cube, n <= 1e4), fp64."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kmvp_oracle
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XSolver
for kernel in ("inverse-distance", "absolute-exponential"):
    for n in (1000, 10000):
        y = kmvp_oracle.uniform_sphere_points(n); b = np.random.RandomState(n).randn(n, 1)
        a = kmvp_oracle.product(kernel=kernel, source_points=y, source_signal=b)
        sol = MI355XSolver(kernel=kernel, dimension=3, precision="float64", rtol=1e-8, maxit=20000)
        sol.prepare_data(source_points=y); sol.prepare_query(target_signal=a)
        sol.query()
        t0 = time.perf_counter(); sol.query(); t = time.perf_counter() - t0
        info = sol.get_additional()
        print(f"{kernel:21s} n={n:6d}: {t*1e3:8.2f} ms, {info['cg_iterations']} iterations, {t/max(info['cg_iterations'],1)*1e6:.0f} us per iteration, residual {info['cg_relative_residual']:.1e}", flush=True)
        sol.done()
