"""Dataset files: schema, synthetic generators, ground truth.

Counterpart of the reference's ``kernel_matrix_benchmarks/datasets.py``:
 * file schema of :1-70 (four float64 arrays + seven attributes) -- ``write_dataset``;
 * ``uniform_cube`` (:248-282: ``seed(n+D)``, ``rand(n,D)*radius``, ``randn(n,1)``) and
   ``uniform_sphere`` (:200-244: golden-angle spiral; the reference's signal is
   unseeded there, here it is seeded so that files are reproducible);
 * dataset names follow ``{task}-{label}-D{D}-E{E}-M{M}-N{N}-{kernel}`` (algos.yaml:38).

Two deliberate differences.  (1) Nothing is ever downloaded (the reference tries
``kernel-matrix-benchmarks.com`` first, :107-109).  (2) The ground truth
``target_signal`` is NOT computed with a dense N x M matrix: the reference's
``write_output`` (:133-195) needs 8*N*M*(D+1) bytes, which stops at a few 1e4
points.  Here the truth is the float64 HIP product (same arithmetic, fp64 VALU
kernels), which tests pin against the reference's outputs on small shapes.
"""
import math
import os
import re

import numpy as np

from kernel_matrix_benchmarks_amd import storage

NAME_RE = re.compile(
    r"^(?P<task>product|solver|attention)-(?P<label>[a-z0-9]+)-D(?P<D>\d+)-E(?P<E>\d+)"
    r"-M(?P<M>\d+)-N(?P<N>\d+)-(?P<kernel>[a-z-]+)$"
)


def parse_name(name):
    m = NAME_RE.match(name)
    if not m:
        raise ValueError(f"dataset name {name!r} does not follow task-label-D-E-M-N-kernel")
    d = m.groupdict()
    for k in ("D", "E", "M", "N"):
        d[k] = int(d[k])
    return d


def cube_points(n_points, dimension, radius=1.0, E=1):
    """datasets.py:256-266.  E > 1 draws ``randn(n, E)`` (identical stream for E = 1)."""
    rs = np.random.RandomState(n_points + dimension)
    y = radius * rs.rand(n_points, dimension)
    b = rs.randn(n_points, E)
    return y, b


def sphere_points(n_points, radius=1.0):
    """datasets.py:210-225 (dimension 3 only)."""
    i = np.arange(n_points, dtype=np.float64)
    yy = 1.0 - (i / float(n_points - 1)) * 2.0 if n_points > 1 else np.zeros(1)
    ry = np.sqrt(np.maximum(1.0 - yy * yy, 0.0))
    theta = math.pi * (3.0 - math.sqrt(5.0)) * i
    return radius * np.stack([np.cos(theta) * ry, yy, np.sin(theta) * ry], axis=1)


def dataset_path(name, root="data"):
    os.makedirs(root, exist_ok=True)
    return os.path.join(root, name + storage.extension())


def write_dataset(*, filename, task, kernel, source_points, target_points=None,
                  source_signal=None, target_signal, normalize_rows=False,
                  short_description="", description="", point_type="float"):
    """Writes the schema of datasets.py:1-70 / :147-195."""
    with storage.open_file(filename, "w") as f:
        f.attrs["kernel"] = kernel
        f.attrs["task"] = task
        f.attrs["point_type"] = point_type
        f.attrs["normalize_rows"] = bool(normalize_rows)
        f.attrs["short_description"] = short_description
        f.attrs["description"] = description
        f["source_points"] = np.asarray(source_points, dtype=np.float64)
        f["target_points"] = np.asarray(
            source_points if target_points is None else target_points, dtype=np.float64)
        f.attrs["same_points"] = target_points is None
        if source_signal is None:
            f["source_signal"] = np.ones((len(source_points), 1))
            f.attrs["density_estimation"] = True
        else:
            f["source_signal"] = np.asarray(source_signal, dtype=np.float64)
            f.attrs["density_estimation"] = False
        f["target_signal"] = np.asarray(target_signal, dtype=np.float64)


def ground_truth(*, kernel, source_points, target_points=None, source_signal=None,
                 normalize_rows=False, device=0):
    """float64 product on the GPU (see the module docstring)."""
    from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct

    algo = MI355XProduct(kernel=kernel, dimension=source_points.shape[1],
                         normalize_rows=normalize_rows, precision=np.float64, device=device)
    try:
        algo.prepare_data(source_points=source_points,
                          target_points=source_points if target_points is None else target_points,
                          same_points=target_points is None,
                          density_estimation=source_signal is None)
        algo.fit()
        algo.prepare_query(source_signal=source_signal)
        algo.query()
        return algo.get_result()
    finally:
        algo.done()


def generate(name, root="data", device=0):
    """Creates the dataset ``name`` (label ``cube`` or ``sphere``) locally."""
    p = parse_name(name)
    if p["M"] != p["N"]:
        raise ValueError("the synthetic generators produce same-points datasets (M == N)")
    n, D, E = p["M"], p["D"], p["E"]
    if p["label"] == "sphere":
        if D != 3:
            raise ValueError("sphere datasets are three-dimensional")
        y = sphere_points(n)
        b = np.random.RandomState(n + D).randn(n, E)
    elif p["label"] == "cube":
        y, b = cube_points(n, D, E=E)
    else:
        raise ValueError(f"unknown dataset label {p['label']!r}")
    task = p["task"]
    normalize = task == "attention"
    a = ground_truth(kernel=p["kernel"], source_points=y, source_signal=b,
                     normalize_rows=normalize, device=device)
    fn = dataset_path(name, root)
    write_dataset(
        filename=fn, task=task, kernel=p["kernel"], source_points=y, source_signal=b,
        target_signal=a, normalize_rows=normalize,
        short_description=f"{p['label']} (N={n}, D={D})",
        description=f"{task.capitalize()} on the {p['label']}, {p['kernel']} (N={n}, D={D})",
    )
    return fn


def get_dataset(name, root="data", device=0):
    """Returns (open file, D); generates the file when missing.  Never downloads."""
    fn = dataset_path(name, root)
    if not os.path.exists(fn):
        generate(name, root, device)
    f = storage.open_file(fn, "r")
    return f, int(f["source_points"].shape[-1])
