"""The drop-in boundary checked against the REFERENCE's own loader (build container only).

INTEGRATION.md section 1 tells a maintainer to add two things to the reference tree: a one-line stub
module ``kernel_matrix_benchmarks/algorithms/mi355x.py`` and two ``algos.yaml`` entries.  This test takes
exactly those two code blocks OUT OF INTEGRATION.md, places them in a temporary overlay (the reference
tree is read-only and nothing of it is copied), and lets the reference's own
``definitions.get_definitions`` / ``algorithm_status`` / ``instantiate_algorithm``
(/root/reference/kernel_matrix_benchmarks/definitions.py:29-44,53-64,90-168) load the plugin -- the code
path of ``run.py --local`` up to the constructor.  Skipped where /root/reference does not exist (the GPU
box): the reference never travels.
"""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "kernel_matrix_benchmarks")),
                                reason="the reference tree is not present on this machine")


def integration_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text.split("## 1.")[1].split("## 2.")[0]
    stub = re.search(r"```python\n(.*?)```", section, flags=re.S).group(1)
    yaml_text = re.search(r"```yaml\n(.*?)```", section, flags=re.S).group(1)
    return stub, yaml_text


CHILD = r"""
import importlib, json, os, sys
overlay, reference, root = sys.argv[1:4]
sys.path[:0] = [reference, root]
import kernel_matrix_benchmarks.algorithms as ref_algorithms
ref_algorithms.__path__.append(os.path.join(overlay, "algorithms"))   # where the maintainer's stub would live
from kernel_matrix_benchmarks import definitions as D
from kernel_matrix_benchmarks.algorithms import base as ref_base

out = {}
yaml_file = os.path.join(overlay, "algos.yaml")
prod = D.get_definitions(definition_file=yaml_file, dimension=3, dataset="product-cube-D3-E1-M1000000-N1000000-gaussian",
                         task="product", hardware="GPU", kernel="gaussian", normalize_rows=False)
att = D.get_definitions(definition_file=yaml_file, dimension=3, dataset="product-cube-D3-E1-M1000-N1000-absolute-exponential",
                        task="attention", hardware="GPU", kernel="absolute-exponential", normalize_rows=True)
sol = D.get_definitions(definition_file=yaml_file, dimension=3, dataset="solver-cube-D3-E1-M100000-N100000-gaussian",
                        task="solver", hardware="GPU", kernel="gaussian", normalize_rows=False)
cpu = D.get_definitions(definition_file=yaml_file, dimension=3, dataset="product-cube-D3-E1-M1000-N1000-gaussian",
                        task="product", hardware="CPU", kernel="gaussian", normalize_rows=False)
out["counts"] = [len(prod), len(att), len(sol), len(cpu)]
out["status"] = [D.algorithm_status(d).name for d in prod + att + sol]
out["modules"] = sorted({d.module for d in prod + att + sol})
algos = [D.instantiate_algorithm(d) for d in prod + att + sol]
import kernel_matrix_benchmarks_amd.algorithms.base as our_base
out["using_reference_bases"] = bool(our_base.USING_REFERENCE_BASES)
out["same_class_objects"] = our_base.BaseProduct is ref_base.BaseProduct and our_base.BaseSolver is ref_base.BaseSolver
out["instances"] = [dict(cls=type(a).__name__, module=type(a).__module__, name=a.name, str=str(a), task=a.task,
                         kernel=a.kernel, dimension=a.dimension, normalize_rows=bool(a.normalize_rows),
                         precision=str(a.precision),
                         is_ref_product=isinstance(a, ref_base.BaseProduct), is_ref_solver=isinstance(a, ref_base.BaseSolver),
                         additional=a.get_additional(), memory=a.get_memory_usage())
                    for a in algos]
for a in algos:
    a.done()
# the reference's failure mode for an unknown kernel (bruteforce.py:82-85): NotImplementedError out of the ctor
bad = prod[0]._replace(arguments=dict(prod[0].arguments, kernel="laplacian"))
try:
    D.instantiate_algorithm(bad)
    out["unknown_kernel"] = "accepted"
except NotImplementedError:
    out["unknown_kernel"] = "NotImplementedError"
print("RESULT " + json.dumps(out))
"""


def test_reference_loader_instantiates_the_plugin_from_the_integration_stub(tmp_path):
    stub, yaml_text = integration_blocks()
    assert "MI355XProduct" in stub and "mi355x-product" in yaml_text and "mi355x-solver" in yaml_text
    overlay = tmp_path / "overlay"
    (overlay / "algorithms").mkdir(parents=True)
    (overlay / "algorithms" / "mi355x.py").write_text(stub)
    (overlay / "algos.yaml").write_text(yaml_text)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")  # the reference tree is read-only
    env.pop("PYTHONPATH", None)
    run = subprocess.run([sys.executable, "-c", CHILD, str(overlay), REFERENCE, ROOT], capture_output=True, text=True,
                         env=env, cwd=str(tmp_path), timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    out = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])

    assert out["counts"] == [3, 3, 1, 0]  # float16 / 32 / 64 products, the same as attention, one solver; nothing on CPU
    assert out["status"] == ["AVAILABLE"] * 7
    assert out["modules"] == ["kernel_matrix_benchmarks.algorithms.mi355x"]  # loaded THROUGH the stub
    assert out["using_reference_bases"] and out["same_class_objects"]
    assert out["unknown_kernel"] == "NotImplementedError"
    p16, p32, p64, a16, a32, a64, s64 = out["instances"]
    for inst, name, precision, kernel, norm in ((p16, "MI355XProduct(float16)", "float16", "gaussian", False),
                                                (a16, "MI355XProduct(float16)", "float16", "absolute-exponential", True),
                                                (p32, "MI355XProduct(float32)", "float32", "gaussian", False),
                                                (p64, "MI355XProduct(float64)", "float64", "gaussian", False),
                                                (a32, "MI355XProduct(float32)", "float32", "absolute-exponential", True),
                                                (a64, "MI355XProduct(float64)", "float64", "absolute-exponential", True)):
        assert inst["cls"] == "MI355XProduct" and inst["module"] == "kernel_matrix_benchmarks_amd.algorithms.mi355x"
        assert inst["name"] == inst["str"] == name and inst["task"] == "product"
        assert (inst["kernel"], inst["dimension"], inst["normalize_rows"], inst["precision"]) == (kernel, 3, norm, precision)
        assert inst["is_ref_product"] and not inst["is_ref_solver"]
        assert inst["additional"] == {} and inst["memory"] == 0.0  # no GPU work before prepare_data (main.py:262-308)
    assert s64["cls"] == "MI355XSolver" and s64["is_ref_solver"] and not s64["is_ref_product"]
    assert s64["task"] == "solver" and s64["name"] == "MI355XSolver(float64, cg, rtol=1e-06)"
