"""Generates ``tests/golden/expected.npz`` by RUNNING THE REFERENCE.

Run once in the build container (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/make_golden.py

It imports ``kernel_matrix_benchmarks.algorithms.bruteforce`` from
``/root/reference`` (read-only), drives ``BruteForceProductBLAS`` /
``BruteForceSolverLAPACK`` through the runner's call order
(runner.py:70-148: prepare_data, fit, prepare_query, query, get_result) on
the seeded inputs of ``golden_cases.py`` and stores only the OUTPUTS.  While
doing so it asserts that ``oracle/kmvp_oracle.py`` reproduces every output --
this is what pins the oracle.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from kernel_matrix_benchmarks.algorithms.bruteforce import (  # noqa: E402  (the reference)
    BruteForceProductBLAS,
    BruteForceSolverLAPACK,
)
import golden_cases  # noqa: E402
import kmvp_oracle  # noqa: E402


def run_reference(case, y, x, b, precision, fast):
    algo = BruteForceProductBLAS(
        kernel=case["kernel"], dimension=case["D"], normalize_rows=case["normalize_rows"],
        precision=precision, fast_sqdists=fast,
    )
    algo.prepare_data(
        source_points=y, target_points=(y if x is None else x),
        same_points=case["same_points"], density_estimation=case["density_estimation"],
    )
    algo.fit()
    algo.prepare_query(source_signal=(np.ones((len(y), 1)) if b is None else b))
    algo.query()
    return algo.get_result()


def main():
    out = {}
    worst = 0.0
    warnings.simplefilter("ignore", RuntimeWarning)  # 1/sqrt(0) = inf is the reference's behaviour
    for case in golden_cases.product_cases():
        y, x, b = golden_cases.make_inputs(case)
        for tag, precision, fast in (
            ("f64", np.float64, False), ("f64fast", np.float64, True),
            ("f32", np.float32, False), ("f32fast", np.float32, True),
        ):
            ref = run_reference(case, y, x, b, precision, fast)
            mine = kmvp_oracle.product(
                kernel=case["kernel"], source_points=y, target_points=x, source_signal=b,
                normalize_rows=case["normalize_rows"], density_estimation=case["density_estimation"],
                precision=precision, fast_sqdists=fast, block_rows=37,
            )
            assert ref.shape == mine.shape, (case["name"], tag, ref.shape, mine.shape)
            fin = np.isfinite(ref)
            assert np.array_equal(fin, np.isfinite(mine)), (case["name"], tag, "inf/nan pattern")
            tol = 1e-12 if precision is np.float64 else 2e-4
            if fast and case["kernel"] != "gaussian":
                tol = max(tol, 1e-6 if precision is np.float64 else 5e-2)  # sqrt near 0 amplifies BLAS order
            scale = max(1.0, float(np.max(np.abs(ref[fin])))) if fin.any() else 1.0
            err = float(np.max(np.abs(ref[fin] - mine[fin]))) / scale if fin.any() else 0.0
            assert err <= tol, (case["name"], tag, err)
            if precision is np.float64 and not fast:
                worst = max(worst, err)
            out[f"{case['name']}/{tag}"] = ref if precision is np.float64 else ref.astype(np.float32)
    print(f"{len(golden_cases.product_cases())} product cases; oracle vs reference (f64, slow) worst rel err {worst:.2e}")

    for case in golden_cases.solver_cases():
        y, b_true = golden_cases.make_solver_inputs(case)
        prod = BruteForceProductBLAS(kernel=case["kernel"], dimension=case["D"])
        prod.prepare_data(source_points=y, target_points=y)
        prod.fit()
        prod.prepare_query(source_signal=b_true)
        prod.query()
        a = prod.get_result()
        sol = BruteForceSolverLAPACK(kernel=case["kernel"], dimension=case["D"])
        sol.prepare_data(source_points=y)
        sol.fit()
        sol.prepare_query(target_signal=a)
        sol.query()
        b_ls = sol.get_result()
        mine = kmvp_oracle.solve(kernel=case["kernel"], source_points=y, target_signal=a)
        res_ref = float(np.linalg.norm(prod.K_ij @ b_ls - a) / np.linalg.norm(a))
        res_mine = kmvp_oracle.relative_residual(
            kernel=case["kernel"], source_points=y, solution=mine, target_signal=a
        )
        assert res_ref < 1e-9 and res_mine < 1e-9, (case["name"], res_ref, res_mine)
        out[f"{case['name']}/a"] = a
        out[f"{case['name']}/b_lstsq"] = b_ls
        out[f"{case['name']}/residual"] = np.array(res_ref)
        print(case["name"], "lstsq residual", res_ref, "max|b-b_true|", float(np.max(np.abs(b_ls - b_true))))

    path = os.path.join(HERE, "golden", "expected.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
