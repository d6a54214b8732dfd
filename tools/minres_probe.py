"""MINRES on the reference's sphere dataset shape (inverse-distance, n points on the unit sphere): iterations, seconds and
microseconds per iteration for float64 (rtol 1e-6) and float32 (rtol 1e-4).  usage: python tools/minres_probe.py [n=10000]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XSolver
import kmvp_oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
y = kmvp_oracle.uniform_sphere_points(n)
rs = np.random.RandomState(n)
b = rs.randn(n, 1)
a = kmvp_oracle.product(kernel="inverse-distance", source_points=y, source_signal=b)
for precision, rtol in (("float64", 1e-6), ("float32", 1e-4)):
    s = MI355XSolver(kernel="inverse-distance", dimension=3, precision=precision, rtol=rtol, maxit=20000)
    s.prepare_data(source_points=y); s.fit(); s.prepare_query(target_signal=a)
    t0 = time.time(); s.query(); dt = time.time() - t0
    d = s.get_additional()
    print(precision, "iterations", d["cg_iterations"], "seconds", round(dt, 4), "us/iter", round(dt / max(d["cg_iterations"], 1) * 1e6, 1), d["device_kernel"], d["cg_relative_residual"])
    s.done()
