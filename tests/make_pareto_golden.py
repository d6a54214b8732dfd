"""Generates ``tests/golden/pareto.json`` by RUNNING THE REFERENCE's Pareto-front rule.

Run once in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/make_pareto_golden.py

Imports ``kernel_matrix_benchmarks.plotting.utils.create_pointset`` from ``/root/reference`` (read-only)
and applies it, for the reference's default axes (plot.py:109-128: x = total-time, y = rmse-error) and two
other metric pairs, to seeded point sets that include ties in y, ties in x, exact duplicates, zeros and a
single point.  Only the inputs' seeds and the OUTPUT (labels of "all" and of "front", in order) are stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from kernel_matrix_benchmarks.plotting.utils import create_pointset  # noqa: E402  (the reference)


def point_set(seed, n, ties):
    """(label, x, y) triples; ``ties`` quantises the values so that equal x / equal y occur."""
    rs = np.random.RandomState(seed)
    x = rs.lognormal(size=n)
    y = rs.lognormal(size=n) * 1e-3
    if ties:
        x = np.round(x * ties) / ties
        y = np.round(y * 1e3 * ties) / (1e3 * ties)
    return [(f"p{i}", float(x[i]), float(y[i])) for i in range(n)]


CASES = [dict(seed=1, n=1, ties=0), dict(seed=2, n=7, ties=0), dict(seed=3, n=40, ties=0), dict(seed=4, n=40, ties=2),
         dict(seed=5, n=60, ties=1), dict(seed=6, n=200, ties=4)]
AXES = [("total-time", "rmse-error"), ("query-time", "max-error"), ("memory-footprint", "median-error")]


def main():
    out = []
    for case in CASES:
        pts = point_set(**case)
        for x_name, y_name in AXES:
            data = [("algo", label, xv, yv) for label, xv, yv in pts]
            ref = create_pointset(data=data, x_name=x_name, y_name=y_name)
            out.append(dict(case, x_name=x_name, y_name=y_name, all=ref["all"]["labels"], front=ref["front"]["labels"],
                            front_x=ref["front"]["x"], front_y=ref["front"]["y"]))
    with open(os.path.join(HERE, "golden", "pareto.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(f"{len(out)} fronts written")


if __name__ == "__main__":
    main()
