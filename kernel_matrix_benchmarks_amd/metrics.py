"""Error / time metrics, counterpart of ``kernel_matrix_benchmarks/plotting/metrics.py``.

``result_errors`` restates metrics.py:36-61 (per-row L2 norm over the E columns of
``error``, then max / mean / median / rmse); the time metrics restate :67-81.
``max-error`` is the "max |err|" of BASELINE.json's metric.
"""
import numpy as np


def result_errors(error):
    norms = np.sqrt(np.sum(np.asarray(error, dtype=np.float64) ** 2, axis=-1))
    return {
        "max": float(np.max(norms)),
        "mean": float(np.mean(norms)),
        "median": float(np.median(norms)),
        "rmse": float(np.sqrt(np.mean(norms ** 2))),
    }


def relative_max_error(result, truth):
    """max_i ||result_i - truth_i|| / max_i ||truth_i|| -- the tolerance the parity
    tests state (the reference defines no threshold)."""
    result = np.asarray(result, dtype=np.float64)
    truth = np.asarray(truth, dtype=np.float64)
    scale = np.max(np.sqrt(np.sum(truth ** 2, axis=-1)))
    return result_errors(result - truth)["max"] / (scale if scale > 0 else 1.0)


def build_time(properties):
    return properties["build_time"]


def query_time(properties):
    return properties["query_time"]


def total_time(properties):
    return properties["build_time"] + properties["query_time"]


def memory_footprint(properties):
    return properties.get("memory_footprint", 0)


def pairs_per_second(properties, n_targets, n_sources):
    """N*M / (build + query): the headline throughput."""
    return float(n_targets) * float(n_sources) / total_time(properties)


ALL_METRICS = {
    "max-error": lambda error, properties: result_errors(error)["max"],
    "mean-error": lambda error, properties: result_errors(error)["mean"],
    "median-error": lambda error, properties: result_errors(error)["median"],
    "rmse-error": lambda error, properties: result_errors(error)["rmse"],
    "build-time": lambda error, properties: build_time(properties),
    "query-time": lambda error, properties: query_time(properties),
    "total-time": lambda error, properties: total_time(properties),
    "memory-footprint": lambda error, properties: memory_footprint(properties),
}


# every metric of the reference is "lower is better" (worst = +inf, metrics.py:87-128)
def pareto_front(points):
    """Pareto frontier of (label, x, y) triples, both metrics lower-is-better.

    Restates ``plotting/utils.create_pointset`` (utils.py:15-76): sort from the best to
    the worst y (ties on x), sweep, keep every point whose x beats the best x seen so far.
    Returns ``{"front": {"x", "y", "labels"}, "all": {...}}`` like the reference."""
    data = sorted(points, key=lambda t: (t[2], t[1]))
    out = {"front": {"x": [], "y": [], "labels": []}, "all": {"x": [], "y": [], "labels": []}}
    best_x = float("inf")
    for label, xv, yv in data:
        out["all"]["x"].append(xv)
        out["all"]["y"].append(yv)
        out["all"]["labels"].append(label)
        if xv < best_x:
            best_x = xv
            out["front"]["x"].append(xv)
            out["front"]["y"].append(yv)
            out["front"]["labels"].append(label)
    return out


def summarize_results(dataset, x_name="total-time", y_name="rmse-error", root="results"):
    """(label, x, y) for every stored run of ``dataset`` and the Pareto front, with the
    reference's default axes (plot.py:109-128)."""
    from kernel_matrix_benchmarks_amd import results as _results

    pts = []
    for props, f in _results.load_all_results(dataset, root=root):
        err = np.asarray(f["error"][:])
        pts.append((props.get("name", "?"), ALL_METRICS[x_name](err, props), ALL_METRICS[y_name](err, props)))
    return pts, pareto_front(pts)
