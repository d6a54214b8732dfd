"""Generates ``tests/golden/expected_f16.npz`` by RUNNING THE REFERENCE in its float16 mode.

    PYTHONDONTWRITEBYTECODE=1 python tests/make_golden_f16.py

The reference's registry runs ``BruteForceProductBLAS`` with ``precision: "float16"`` (algos.yaml:157,160): the
points and the signal are cast to float16 (bruteforce.py:103-111,126), the kernel matrix and the product are formed
in float16 by numpy, the result is returned as float64 (base.py:107-116).  This script drives exactly that on the
seeded product cases of ``golden_cases.py`` (slow squared-distance form) and stores the OUTPUTS.  They pin the
plugin's ``precision="float16"`` mode (float16-rounded inputs, float32 arithmetic on the GPU): it has to be at least
as close to the float64 truth as the reference's own float16 run is.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from kernel_matrix_benchmarks.algorithms.bruteforce import BruteForceProductBLAS  # noqa: E402  (the reference)
import golden_cases  # noqa: E402


def main():
    out = {}
    warnings.simplefilter("ignore", RuntimeWarning)  # overflow / divide in float16 are the reference's behaviour
    for case in golden_cases.product_cases():
        y, x, b = golden_cases.make_inputs(case)
        algo = BruteForceProductBLAS(kernel=case["kernel"], dimension=case["D"], normalize_rows=case["normalize_rows"],
                                     precision=np.float16, fast_sqdists=False)
        algo.prepare_data(source_points=y, target_points=(y if x is None else x), same_points=case["same_points"],
                          density_estimation=case["density_estimation"])
        algo.fit()
        algo.prepare_query(source_signal=(np.ones((len(y), 1)) if b is None else b))
        algo.query()
        ref = algo.get_result()
        assert ref.dtype == np.float64
        out[f"{case['name']}/f16"] = ref.astype(np.float32)  # float16 values are exactly representable
    path = os.path.join(HERE, "golden", "expected_f16.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
