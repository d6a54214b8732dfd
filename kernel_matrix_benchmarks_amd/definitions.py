"""``algos.yaml`` -> experiment definitions, counterpart of
``kernel_matrix_benchmarks/definitions.py``.

Same registry format (algos.yaml:5-117) and the same expansion rules as
``get_definitions`` (definitions.py:90-168): entries are filtered on ``disabled``,
``hardware`` and the task flag, run-groups on ``fnmatch`` of the dataset name, and
every ``args`` dict is merged over ``{kernel, dimension, normalize_rows}``.
``instantiate_algorithm`` = ``importlib`` + ``constructor(**arguments)``
(definitions.py:29-44).
"""
import collections
import fnmatch
import importlib
import os

import yaml

Definition = collections.namedtuple(
    "Definition",
    ["algorithm", "constructor", "module", "docker_tag", "arguments", "query_argument_groups"],
)

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "algos.yaml")


def load_registry(definition_file=DEFAULT_FILE):
    with open(definition_file, "r") as f:
        return yaml.load(f, yaml.SafeLoader)


def instantiate_algorithm(definition):
    module = importlib.import_module(definition.module)
    return getattr(module, definition.constructor)(**definition.arguments)


def algorithm_available(definition):
    """True when the module imports and has the constructor (definitions.py:53-62)."""
    try:
        return hasattr(importlib.import_module(definition.module), definition.constructor)
    except ImportError:
        return False


def get_definitions(definition_file=DEFAULT_FILE, dimension=3,
                    dataset="product-cube-D3-E1-M1000-N1000-gaussian", task="product",
                    hardware="GPU", kernel="gaussian", normalize_rows=False, run_disabled=False):
    out = []
    for name, algo in load_registry(definition_file).items():
        if algo.get("disabled", False) and not run_disabled:
            continue
        if algo.get("hardware", "CPU") != hardware or not algo.get(task, False):
            continue
        for key in ("docker-tag", "module", "constructor"):
            if key not in algo:
                raise Exception(f'algorithm {name} does not define a "{key}" property')
        for group_name, group in algo["run-groups"].items():
            if "datasets" not in group:
                raise ValueError(f'The field "datasets" is missing for run-group "{group_name}" of algo "{name}".')
            if not any(fnmatch.fnmatch(dataset, pattern) for pattern in group["datasets"]):
                continue
            for args in group.get("args", [{}]):
                merged = dict({"kernel": kernel, "dimension": dimension,
                               "normalize_rows": normalize_rows}, **args)
                out.append(Definition(
                    algorithm=name, docker_tag=algo["docker-tag"], module=algo["module"],
                    constructor=algo["constructor"], arguments=merged,
                    query_argument_groups=group.get("query-args", [{}]),
                ))
    return out
