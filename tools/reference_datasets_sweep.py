"""The reference's OWN dataset list (datasets.py:382-414: product / solver on the sphere (inverse-distance) and on the
"cube" (gaussian), n = 1000, 2000, 5000, 10000) through the plugin's runner protocol: build time, best query time and the
error / residual each definition of algos.yaml reaches.  What a user of the reference sees after switching.

    python tools/reference_datasets_sweep.py [--root /tmp/kmb_data] [--sizes 1000,2000,5000,10000]
"""
import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--root", default=None)
    p.add_argument("--sizes", default="1000,2000,5000,10000")
    a = p.parse_args()
    from kernel_matrix_benchmarks_amd import runner

    root = a.root or tempfile.mkdtemp(prefix="kmb_")
    names = []
    for n in [int(v) for v in a.sizes.split(",")]:
        names += [f"product-sphere-D3-E1-M{n}-N{n}-inverse-distance", f"product-cube-D3-E1-M{n}-N{n}-gaussian",
                  f"solver-sphere-D3-E1-M{n}-N{n}-inverse-distance", f"solver-cube-D3-E1-M{n}-N{n}-gaussian"]
    for name in names:
        res = runner.run_dataset(name, data_root=os.path.join(root, "data"), results_root=os.path.join(root, "results"),
                                 verbose=False)
        for fn, attrs, result in res:
            import numpy as np
            from kernel_matrix_benchmarks_amd import hdf5_lite  # noqa: F401  (results were stored through it)

            keep = {k: attrs[k] for k in ("name", "build_time", "query_time", "cg_iterations", "cg_relative_residual", "cg_converged",
                                          "device_kernel") if k in attrs}
            keep = {k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in keep.items()}
            print(name, keep, flush=True)

if __name__ == "__main__":
    main()
