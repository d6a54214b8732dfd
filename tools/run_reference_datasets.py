"""Runs the plugins through the harness on the reference's own dataset family
(datasets.py:383-427: product/solver x sphere x {inverse-distance, gaussian}, n = 1000 .. 10000;
the reference's "*-cube-*" names are sphere points too, datasets.py:401,409 -- generated here with
the sphere generator under the sphere label).  Prints build/query times and error metrics."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import metrics, runner, storage

sizes = [int(a) for a in sys.argv[1:]] or [1000, 10000]
tmp = tempfile.mkdtemp()
for n in sizes:
    for task in ("product", "solver"):
        for kernel in ("inverse-distance", "gaussian"):
            name = f"{task}-sphere-D3-E1-M{n}-N{n}-{kernel}"
            stored = runner.run_dataset(name, hardware="GPU", runs=2, data_root=os.path.join(tmp, "data"),
                                        results_root=os.path.join(tmp, "results"), verbose=False)
            for fn, attrs, result in stored:
                f = storage.open_file(fn, "r")
                err = np.asarray(f["error"][:]); f.close()
                extra = ""
                if task == "solver":
                    extra = f" iters={attrs['cg_iterations']} residual={attrs['cg_relative_residual']:.1e}"
                print(f"{name:55s} {attrs['name']:42s} build {attrs['build_time']:.1e}s query {attrs['query_time']:.3e}s "
                      f"max-error {metrics.result_errors(err)['max']:.2e}{extra}", flush=True)
