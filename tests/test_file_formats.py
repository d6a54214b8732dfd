"""The dataset and result FILES are the reference's formats, checked with the HDF5
command-line tools instead of with the module that wrote them.

Reference schemas: datasets -- kernel_matrix_benchmarks/datasets.py:1-70 and write_output :133-195
(four float64 arrays, eight attributes: four strings, three booleans ... written through h5py);
results -- results.py:1-48,96-123 (arrays ``result`` and ``error``, the runner's attributes
runner.py:151-163).  h5py stores ``str`` as a variable-length UTF-8 string scalar, numpy booleans as
an int8 enum {FALSE, TRUE}, Python floats / ints as 64-bit scalars; that is what ``h5dump -H`` must show.
"""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from kernel_matrix_benchmarks_amd import datasets, definitions, results, storage

H5DUMP = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)

pytestmark = pytest.mark.skipif(H5DUMP is None or storage.backend() == "npz",
                                reason="needs the h5dump tool and an HDF5 backend")


def header(path):
    """{"attributes": {name: type text}, "datasets": {name: (type, shape)}} from ``h5dump -H``."""
    text = subprocess.run([H5DUMP, "-H", path], capture_output=True, text=True, check=True).stdout
    attrs, dsets = {}, {}
    for m in re.finditer(r'ATTRIBUTE "([^"]+)" \{\s*DATATYPE\s+(.*?)\s*DATASPACE\s+SCALAR', text, flags=re.S):
        attrs[m.group(1)] = " ".join(m.group(2).split())
    for m in re.finditer(r'DATASET "([^"]+)" \{\s*DATATYPE\s+(\S+)\s*DATASPACE\s+SIMPLE \{ \( ([0-9, ]+) \)', text):
        dsets[m.group(1)] = (m.group(2), tuple(int(v) for v in m.group(3).split(",")))
    return {"attributes": attrs, "datasets": dsets}


def is_h5py_str(t):
    return t.startswith("H5T_STRING") and "STRSIZE H5T_VARIABLE" in t and "CSET H5T_CSET_UTF8" in t


def is_h5py_bool(t):
    return t.startswith("H5T_ENUM") and "H5T_STD_I8LE" in t and '"FALSE" 0' in t and '"TRUE" 1' in t


def test_dataset_file_schema(tmp_path):
    y, b = datasets.cube_points(37, 3)
    fn = datasets.dataset_path("product-cube-D3-E1-M37-N37-gaussian", str(tmp_path))
    datasets.write_dataset(filename=fn, task="product", kernel="gaussian", source_points=y, source_signal=b,
                           target_signal=2 * b, short_description="cube (N=37, D=3)", description="Product on the cube")
    assert fn.endswith(".hdf5")
    h = header(fn)
    # datasets.py:147-195
    assert h["datasets"] == {"source_points": ("H5T_IEEE_F64LE", (37, 3)), "target_points": ("H5T_IEEE_F64LE", (37, 3)),
                             "source_signal": ("H5T_IEEE_F64LE", (37, 1)), "target_signal": ("H5T_IEEE_F64LE", (37, 1))}
    assert sorted(h["attributes"]) == sorted(["kernel", "task", "point_type", "normalize_rows", "short_description",
                                              "description", "same_points", "density_estimation"])
    for k in ("kernel", "task", "point_type", "short_description", "description"):
        assert is_h5py_str(h["attributes"][k]), (k, h["attributes"][k])
    for k in ("normalize_rows", "same_points", "density_estimation"):
        assert is_h5py_bool(h["attributes"][k]), (k, h["attributes"][k])
    # density estimation: source_signal = ones((M, 1)), flag set (datasets.py:177-180)
    fn2 = str(tmp_path / "density.hdf5")
    datasets.write_dataset(filename=fn2, task="product", kernel="gaussian", source_points=y, target_points=y[:5],
                           source_signal=None, target_signal=np.ones((5, 1)))
    f = storage.open_file(fn2, "r")
    try:
        assert bool(f.attrs["density_estimation"]) is True and bool(f.attrs["same_points"]) is False
        assert np.array_equal(np.asarray(f["source_signal"][:]), np.ones((37, 1)))
        assert np.asarray(f["target_points"][:]).shape == (5, 3)
    finally:
        f.close()


def test_result_file_schema(tmp_path):
    d = definitions.Definition("mi355x-product", "MI355XProduct", "kernel_matrix_benchmarks_amd.algorithms.mi355x", "tag",
                               {"kernel": "gaussian", "dimension": 3, "normalize_rows": False, "precision": "float32"}, [{}])
    attrs = {  # runner.py:151-163 + get_additional()
        "dataset": "product-cube-D3-E1-M37-N37-gaussian", "algo": "mi355x-product", "name": "MI355XProduct(float32)",
        "kernel": "gaussian", "run_count": 3, "build_time": 0.004, "query_time": 0.041, "memory_footprint": 1234.5,
        "device_kernel_ms": 40.5, "device_kernel": "cell_kernel", "n_gpus": 1,
    }
    res = np.arange(37.0).reshape(37, 1)
    fn = results.store_result(dataset=attrs["dataset"], definition=d, query_arguments={}, attrs=attrs, result=res,
                              error=res * 1e-7, root=str(tmp_path))
    # results.py:73-93: results/<dataset>/<algo>/<args>.hdf5
    assert fn == os.path.join(str(tmp_path), attrs["dataset"], "mi355x-product",
                              "dimension_3_kernel_gaussian_normalize_rows_false_precision_float32.hdf5")
    h = header(fn)
    assert h["datasets"] == {"result": ("H5T_IEEE_F64LE", (37, 1)), "error": ("H5T_IEEE_F64LE", (37, 1))}
    assert sorted(h["attributes"]) == sorted(attrs)
    for k in ("dataset", "algo", "name", "kernel", "device_kernel"):
        assert is_h5py_str(h["attributes"][k]), k
    for k in ("build_time", "query_time", "memory_footprint", "device_kernel_ms"):
        assert h["attributes"][k] == "H5T_IEEE_F64LE", k
    for k in ("run_count", "n_gpus"):
        assert h["attributes"][k] == "H5T_STD_I64LE", k
    (props, f), = list(results.load_all_results(attrs["dataset"], root=str(tmp_path)))
    assert props["algo"] == "mi355x-product" and props["run_count"] == 3 and abs(props["query_time"] - 0.041) < 1e-15
