"""Randomised solver sweep: K b = a for random clouds / kernels / precisions / numbers of right-hand sides through
MI355XSolver; whatever the solver reports is checked against the float64 numpy oracle: a solve reported as converged must
have a true residual ||K b - a|| / ||a|| <= 2 rtol (evaluated in float64 on the inputs as the working precision sees them),
and the reported residual must agree with it.  usage: python tools/fuzz_solver.py [cases=150] [seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XSolver  # noqa: E402
import kmvp_oracle  # noqa: E402  (checker only)


def sweep(cases, seed, verbose=True):
    rs = np.random.RandomState(seed)
    failures, stats = [], {"converged": 0, "not_converged": 0}
    for i in range(cases):
        kernel = ["gaussian", "absolute-exponential", "inverse-distance"][rs.randint(3)]
        n = int(rs.choice([40, 200, 1000, 3000, 12000, 40000]))
        D = int(rs.choice([1, 2, 3, 3, 5]))
        E = int(rs.choice([1, 1, 3]))
        precision = ["float32", "float64"][rs.randint(2)]
        rtol = float(rs.choice([1e-3, 1e-5, 1e-8])) if precision == "float64" else float(rs.choice([1e-2, 1e-4]))
        refine = "float32" if (precision == "float64" and kernel == "gaussian" and rs.rand() < 0.3) else None
        # a spread that keeps the Gaussian matrix solvable: points a few kernel widths apart
        spread = float(rs.choice([3.0, 10.0, 30.0])) * n ** (1.0 / D) / 10.0
        wp = np.float32 if precision == "float32" else np.float64
        if kernel == "inverse-distance":
            y = rs.randn(n, 3)
            y /= np.linalg.norm(y, axis=1, keepdims=True)  # the reference's solver-sphere datasets
            D = 3
        else:
            y = rs.rand(n, D) * spread
        y = y.astype(wp).astype(np.float64)
        a = rs.randn(n, E).astype(wp).astype(np.float64)
        c = dict(kernel=kernel, n=n, D=D, E=E, precision=precision, rtol=rtol, refine=refine, spread=round(spread, 2))
        try:
            algo = MI355XSolver(kernel=kernel, dimension=D, precision=precision, rtol=rtol, maxit=600, refine=refine)
            try:
                algo.prepare_data(source_points=y)
                algo.fit()
                algo.prepare_query(target_signal=a)
                algo.query()
                b = algo.get_result()
                info = algo.get_additional()
            finally:
                algo.done()
        except Exception as exc:  # noqa: BLE001
            failures.append(f"case {i} {c}: {type(exc).__name__}: {exc}")
            continue
        ok = bool(info["cg_converged"])
        stats["converged" if ok else "not_converged"] += 1
        msg = None
        if b.shape != (n, E):
            msg = f"shape {b.shape}"
        elif ok:
            rows = None if n <= 3000 else np.sort(rs.choice(n, size=300, replace=False))
            Kb = kmvp_oracle.product(kernel=kernel, source_points=y, source_signal=b, rows=rows)
            ar = a if rows is None else a[rows]
            res = float(np.linalg.norm(Kb - ar) / np.linalg.norm(ar))
            # (a row sample of a residual vector fluctuates: factor 3 there)
            if not np.isfinite(b).all() or res > (2.0 if rows is None else 3.0) * rtol:
                msg = f"reported converged (residual {info['cg_relative_residual']:.2e}) but the true residual is {res:.2e}"
        if msg:
            failures.append(f"case {i} {c}: {msg}")
            if verbose:
                print("FAIL " + failures[-1], flush=True)
        if verbose and (i + 1) % 25 == 0:
            print(f"... {i + 1} cases, {len(failures)} failures, {stats}", flush=True)
    return stats, failures


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    stats, failures = sweep(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n_cases} cases, {len(failures)} failures; {stats}")
    sys.exit(1 if failures else 0)
