"""Result files, counterpart of ``kernel_matrix_benchmarks/results.py``.

Layout ``results/<dataset>/<algo>/<args>.hdf5`` with the file-name rule of
results.py:73-93 (JSON of constructor + query arguments, ``\\W+`` -> ``_``) and the
content of results.py:96-123: arrays ``result`` and ``error`` plus the attributes
the runner collects (runner.py:151-163).
"""
import json
import os
import re

import numpy as np

from kernel_matrix_benchmarks_amd import storage


class _Encoder(json.JSONEncoder):
    def default(self, o):
        if isinstance(o, np.bool_):
            return bool(o)
        if isinstance(o, np.integer):
            return int(o)
        if isinstance(o, np.floating):
            return float(o)
        if isinstance(o, type):
            return o.__name__
        return super().default(o)


def result_filename(dataset=None, definition=None, query_arguments=None, root="results"):
    parts = [root]
    if dataset:
        parts.append(dataset)
    if definition:
        parts.append(definition.algorithm)
        args = dict(definition.arguments, **(query_arguments or {}))
        flat = re.sub(r"\W+", "_", json.dumps(args, sort_keys=True, cls=_Encoder)).strip("_")
        parts.append(flat + storage.extension())
    return os.path.join(*parts)


def store_result(*, dataset, definition, query_arguments, attrs, result, error, root="results"):
    fn = result_filename(dataset, definition, query_arguments, root)
    os.makedirs(os.path.dirname(fn), exist_ok=True)
    with storage.open_file(fn, "w") as f:
        for k, v in attrs.items():
            f.attrs[k] = v
        f["result"] = result
        f["error"] = error
    return fn


def load_all_results(dataset=None, root="results"):
    """Yields (attributes, open file) for every stored result (results.py:126-140)."""
    for d, _, files in os.walk(result_filename(dataset, root=root)):
        for fn in sorted(files):
            if os.path.splitext(fn)[-1] not in (".hdf5", ".npz"):
                continue
            f = storage.open_file(os.path.join(d, fn), "r")
            try:
                yield dict(f.attrs), f
            finally:
                f.close()
