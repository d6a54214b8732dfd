"""A long conjugate-gradient solve (ill-conditioned Gaussian system, fp64): hipGraph replay vs plain launches."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kmvp_oracle
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XSolver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
y, b = kmvp_oracle.uniform_cube(n, 3)
a = kmvp_oracle.product(kernel="gaussian", source_points=y * 4, source_signal=b)
sol = MI355XSolver(kernel="gaussian", dimension=3, precision="float64", rtol=1e-10, maxit=6000)
sol.prepare_data(source_points=y * 4); sol.prepare_query(target_signal=a)
sol.query()
t0 = time.perf_counter(); sol.query(); t = time.perf_counter() - t0
info = sol.get_additional()
print(f"graph={'off' if os.environ.get('KMVP_NO_GRAPH') else 'on'} n={n}: {t*1e3:.1f} ms, {info['cg_iterations']} iterations, {t/max(info['cg_iterations'],1)*1e6:.1f} us per iteration, residual {info['cg_relative_residual']:.1e}", flush=True)
sol.done()
