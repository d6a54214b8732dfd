"""CPU tests: the C-ABI library loads and exports what include/kmvp.h declares; host
logic (registry, file names, sharding arithmetic, runner protocol, containers).
No compute call is made here -- there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import golden_cases
import kmvp_oracle
from kernel_matrix_benchmarks_amd import _lib, datasets, definitions, metrics, results, runner, sharding, storage
from kernel_matrix_benchmarks_amd.algorithms import base, mi355x

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "kmvp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kmvp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"libkmvp.so lacks {name}"
    assert sorted(s[0] for s in _lib.SYMBOLS) == names, "binding table and header disagree"
    assert lib.kmvp_abi_version() == 1


def test_library_is_the_hip_build():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, "libkmvp.so carries no gfx950 code object"
    assert b"lowd_kernel" in blob


def test_no_product_code_touches_the_oracle():
    pkg = os.path.join(ROOT, "kernel_matrix_benchmarks_amd")
    for d, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(d, fn)).read()
                assert "kmvp_oracle" not in text and "c_oracle" not in text, fn


def test_constructor_contract():
    with pytest.raises(NotImplementedError):  # bruteforce.py:82-85
        mi355x.MI355XProduct(kernel="laplacian", dimension=3)
    with pytest.raises(NotImplementedError):
        mi355x.MI355XSolver(kernel="laplacian", dimension=3)
    # README.md:51-59's attention kernel exp(<x, y>): served through the Gaussian identity
    dot = mi355x.MI355XProduct(kernel="exp-dot", dimension=64, normalize_rows=True, precision="bfloat16")
    assert dot._device_kernel_fn == "gaussian" and dot.kernel == "exp-dot"
    dot.done()
    assert datasets.parse_name("attention-cube-D64-E64-M65536-N65536-exp-dot")["kernel"] == "exp-dot"
    assert mi355x.MI355XSolver(kernel="inverse-distance", dimension=3).method == "minres"
    assert mi355x.MI355XSolver(kernel="gaussian", dimension=3).method == "cg"
    with pytest.raises((NotImplementedError, TypeError)):
        mi355x.MI355XProduct(kernel="gaussian", dimension=3, precision="int8")
    p16 = mi355x.MI355XProduct(kernel="gaussian", dimension=3, precision="float16")  # algos.yaml:157
    assert str(p16) == "MI355XProduct(float16)" and p16._round == np.float16 and p16._host_dtype == np.float32
    p16.done()
    p = mi355x.MI355XProduct(kernel="gaussian", dimension=3, normalize_rows=True, precision="float32")
    assert p.task == "product" and str(p) == "MI355XProduct(float32)" and p.normalize_rows
    assert isinstance(p, base.BaseProduct) and p.get_additional() == {} and p.get_memory_usage() == 0.0
    s = mi355x.MI355XSolver(kernel="gaussian", dimension=3, precision=np.float64)
    assert s.task == "solver" and isinstance(s, base.BaseSolver)
    p.done()
    s.done()


def test_no_gpu_means_a_loud_failure_not_a_fallback():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    p = mi355x.MI355XProduct(kernel="gaussian", dimension=3)
    y = np.random.rand(8, 3)
    with pytest.raises(_lib.KmvpError):
        p.prepare_data(source_points=y, target_points=y, same_points=True)


def test_registry_expansion_rules():
    defs = definitions.get_definitions(dataset="product-cube-D3-E1-M1000-N1000-gaussian")
    assert [d.arguments["precision"] for d in defs] == ["float16", "float32", "float64"]  # the reference's three (algos.yaml:157-162)
    d = defs[1]
    assert d.arguments["kernel"] == "gaussian" and d.arguments["dimension"] == 3
    assert d.arguments["normalize_rows"] is False and d.query_argument_groups == [{}]
    assert definitions.algorithm_available(d)
    assert definitions.get_definitions(hardware="CPU") == []
    sol = definitions.get_definitions(task="solver", dataset="solver-cube-D3-E1-M1000-N1000-gaussian")
    assert [s.constructor for s in sol] == ["MI355XSolver"] * 2
    inv = definitions.get_definitions(task="solver", kernel="inverse-distance",
                                      dataset="solver-sphere-D3-E1-M1000-N1000-inverse-distance")
    assert [d.arguments["kernel"] for d in inv] == ["inverse-distance"] * 2  # the reference's solver datasets
    att = definitions.get_definitions(task="attention", normalize_rows=True)
    assert att and all(a.arguments["normalize_rows"] for a in att)


def test_result_filename_rule():
    d = definitions.Definition("algo-x", "C", "m", "t", {"kernel": "gaussian", "dimension": 3,
                                                         "normalize_rows": np.bool_(False), "precision": "float32"}, [{}])
    fn = results.result_filename("ds", d, {"h": 3}, root="results")
    # results.py:73-93: json.dumps(sorted) with every run of non-word characters -> "_"
    assert fn == os.path.join("results", "ds", "algo-x",
                              "dimension_3_h_3_kernel_gaussian_normalize_rows_false_precision_float32" + storage.extension())


def test_dataset_names_and_generators():
    p = datasets.parse_name("product-cube-D3-E1-M1000000-N1000000-gaussian")
    assert (p["task"], p["label"], p["D"], p["E"], p["M"], p["N"], p["kernel"]) == (
        "product", "cube", 3, 1, 1000000, 1000000, "gaussian")
    p = datasets.parse_name("solver-sphere-D3-E1-M1000-N1000-inverse-distance")
    assert p["kernel"] == "inverse-distance" and p["task"] == "solver"
    with pytest.raises(ValueError):
        datasets.parse_name("glove-25-angular")
    y, b = datasets.cube_points(100, 3)
    yo, bo = kmvp_oracle.uniform_cube(100, 3)
    assert np.array_equal(y, yo) and np.array_equal(b, bo)
    np.testing.assert_allclose(datasets.sphere_points(10), kmvp_oracle.uniform_sphere_points(10), atol=1e-15)
    s = datasets.sphere_points(1000)
    np.testing.assert_allclose(np.linalg.norm(s, axis=1), 1.0, atol=1e-12)


def test_shard_ranges_partition_the_sources():
    for M in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            parts = [sharding.shard_range(M, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == M
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def test_spatial_order_keeps_cells_together_and_sums_unchanged():
    """Sharded Gaussian products hand rank r the r-th slice of the sources in cell order
    (sharding.spatial_order): a permutation, each slice confined to its share of the cells, the
    signal permuted the same way -- the shard sums still add up to the unsharded product."""
    rs = np.random.RandomState(8)
    y = rs.rand(4000, 3)
    b = rs.randn(4000, 1)
    header = open(os.path.join(os.path.dirname(__file__), "..", "kernel_matrix_benchmarks_amd", "csrc", "kmvp_cell.hpp")).read()
    assert float(re.search(r"CELL_T_MAX = ([0-9.]+)f", header).group(1)) == sharding.CELL_T_MAX  # one grid on both sides
    order = sharding.spatial_order(y)
    assert sorted(order.tolist()) == list(range(4000))
    cells = sharding.cell_indices(y.astype(np.float32))
    assert cells.max(axis=0).tolist() == [9, 9, 9]  # unit cube: ten cells of side 0.1 <= 0.103 per axis
    keys = cells[:, 0] + 1024 * cells[:, 1] + 1024 * 1024 * cells[:, 2]
    assert (np.diff(keys[order]) >= 0).all()
    world = 4
    total = np.zeros((50, 1))
    distinct = 0
    for rank in range(world):
        lo, hi = sharding.shard_range(4000, rank, world)
        idx = order[lo:hi]
        distinct += len(np.unique(keys[idx]))
        total += kmvp_oracle.product(kernel="gaussian", source_points=y[idx], target_points=y[:50], source_signal=b[idx])
    assert distinct <= len(np.unique(keys)) + world - 1  # a cell is split over at most two neighbouring ranks
    full = kmvp_oracle.product(kernel="gaussian", source_points=y, target_points=y[:50], source_signal=b)
    np.testing.assert_allclose(total, full, rtol=1e-12, atol=1e-12)
    assert sharding.spatial_order(rs.rand(10, 4)) is None                  # D > 3: no cell grid
    assert sharding.spatial_order(np.array([[0.0, 0, 0], [np.inf, 0, 0]])) is None
    assert sharding.spatial_order(np.array([[0.0, 0, 0], [200.0, 0, 0]])) is None  # > 1024 cells along an axis


def test_plugin_shards_gaussian_sources_cell_by_cell(monkeypatch):
    """MI355XProduct with a 2-rank communicator (device context replaced by a recorder): the Gaussian
    hands the library its slice of the sources in cell order with the signal permuted alike; the
    inverse-distance kernel keeps the caller's order (index-based zero rule)."""
    calls = {}

    class Recorder:
        def __init__(self, device=0):
            pass

        def set_option(self, key, value):
            calls.setdefault("options", {})[key] = value

        def set_points(self, y, x, dtype, j_offset=0, M_total=None):
            calls["points"] = (y.copy(), None if x is None else x.copy(), j_offset, M_total)

        def set_signal(self, b):
            calls["signal"] = None if b is None else b.copy()

    class FakeComm:
        rank, world = 1, 2

        def attach(self, ctx):
            calls["attached"] = True

    monkeypatch.setattr(_lib, "Context", Recorder)
    rs = np.random.RandomState(3)
    y = rs.rand(999, 3)
    b = rs.randn(999, 1)
    lo, hi = sharding.shard_range(999, 1, 2)
    for kernel in ("gaussian", "inverse-distance"):
        algo = mi355x.MI355XProduct(kernel=kernel, dimension=3, normalize_rows=False, precision="float32", comm=FakeComm())
        algo.prepare_data(source_points=y, target_points=y, same_points=True)
        algo.prepare_query(source_signal=b)
        ys, xs, j_offset, m_total = calls["points"]
        order = sharding.spatial_order(y.astype(np.float32)) if kernel == "gaussian" else np.arange(999)
        assert calls["attached"] and (j_offset, m_total) == (lo, 999)
        assert np.array_equal(ys, y.astype(np.float32)[order][lo:hi])
        assert np.array_equal(xs, y.astype(np.float32))  # all targets, caller's order
        assert np.array_equal(calls["signal"], b.astype(np.float32)[order][lo:hi])
    # the solver never reorders: a rank's signal is the slice [lo, hi) of the replicated Krylov vector
    for precision in ("float32", np.float64):
        solver = mi355x.MI355XSolver(kernel="gaussian", dimension=3, normalize_rows=False, precision=precision, comm=FakeComm())
        solver.prepare_data(source_points=y)
        ys, xs, j_offset, m_total = calls["points"]
        dt = np.float32 if precision == "float32" else np.float64
        assert (j_offset, m_total) == (lo, 999)
        assert np.array_equal(ys, y.astype(dt)[lo:hi]) and np.array_equal(xs, y.astype(dt))


def test_container_roundtrip(tmp_path):
    fn = str(tmp_path / ("roundtrip" + storage.extension()))
    a = np.arange(12.0).reshape(4, 3)
    with storage.open_file(fn, "w") as f:
        f["a"] = a
        f.attrs["kernel"] = "gaussian"
        f.attrs["same_points"] = True
        f.attrs["build_time"] = 0.25
        f.attrs["run_count"] = 2
    f = storage.open_file(fn, "r")
    try:
        assert np.array_equal(np.asarray(f["a"][:]), a)
        k = f.attrs["kernel"]
        assert (k.decode() if isinstance(k, bytes) else k) == "gaussian"
        assert bool(f.attrs["same_points"]) is True
        assert float(f.attrs["build_time"]) == 0.25 and int(f.attrs["run_count"]) == 2
    finally:
        f.close()


def test_metrics_definition():
    err = np.array([[3.0, 4.0], [0.0, 0.0], [1.0, 0.0]])
    m = metrics.result_errors(err)
    assert m == kmvp_oracle.result_errors(err)
    props = {"build_time": 1.0, "query_time": 3.0}
    assert metrics.total_time(props) == 4.0 and metrics.pairs_per_second(props, 10, 20) == 50.0
    assert metrics.ALL_METRICS["max-error"](err, props) == 5.0


def test_pareto_front_rule():
    # utils.py:15-76: sweep from the best y, keep points that improve the best x so far
    pts = [("a", 1.0, 5.0), ("b", 2.0, 3.0), ("c", 3.0, 4.0), ("d", 4.0, 1.0), ("e", 0.5, 6.0)]
    out = metrics.pareto_front(pts)
    assert out["all"]["labels"] == ["d", "b", "c", "a", "e"]
    assert out["front"]["labels"] == ["d", "b", "a", "e"]  # c is dominated by b
    assert out["front"]["x"] == [4.0, 2.0, 1.0, 0.5]


def test_pareto_front_matches_the_reference(expected):
    """tests/golden/pareto.json: labels of "all" and "front" returned by the REFERENCE's create_pointset
    (plotting/utils.py:15-76, run by tests/make_pareto_golden.py) on seeded point sets with ties and duplicates."""
    import json

    import make_pareto_golden

    with open(os.path.join(ROOT, "tests", "golden", "pareto.json")) as f:
        golden = json.load(f)
    assert len(golden) == len(make_pareto_golden.CASES) * len(make_pareto_golden.AXES) == 18
    for g in golden:
        pts = make_pareto_golden.point_set(g["seed"], g["n"], g["ties"])
        out = metrics.pareto_front(pts)
        assert out["all"]["labels"] == g["all"], g["seed"]
        assert out["front"]["labels"] == g["front"], g["seed"]
        assert out["front"]["x"] == g["front_x"] and out["front"]["y"] == g["front_y"]


def test_signal_and_point_shapes_are_checked_before_the_library_reads_them(monkeypatch):
    """kmvp_set_signal copies M * E elements from the host pointer, so a signal with fewer rows than
    the points must never reach it (the reference fails in its matmul, bruteforce.py:150)."""
    real = _lib.Context

    class Quiet:
        comm_world = 0

        def __init__(self, device=0):
            pass

        def set_option(self, key, value):
            pass

        def set_points(self, y, x, dtype, j_offset=0, M_total=None):
            pass

    monkeypatch.setattr(_lib, "Context", Quiet)
    y = np.random.RandomState(0).rand(50, 3)
    p = mi355x.MI355XProduct(kernel="gaussian", dimension=3)
    p.prepare_data(source_points=y, target_points=y, same_points=True)
    with pytest.raises(ValueError):
        p.prepare_query(source_signal=np.ones((49, 1)))
    with pytest.raises(ValueError):
        p.prepare_query(source_signal=np.ones((51, 2)))
    s = mi355x.MI355XSolver(kernel="gaussian", dimension=3)
    s.prepare_data(source_points=y)
    with pytest.raises(ValueError):
        s.prepare_query(target_signal=np.ones((49, 1)))
    # the typed wrapper checks too (callers of the C ABI through _lib.Context)
    c = real.__new__(real)  # no GPU: the object is never bound to a kmvp_ctx, the checks come first
    c._ctx = None
    c.M, c.N, c.D = 50, 50, 3
    with pytest.raises(ValueError):
        real.set_signal(c, np.ones((49, 1), dtype=np.float32))
    with pytest.raises(ValueError):
        real.cg_solve(c, "gaussian", np.ones((49, 1)), 1e-6, 10)


class OracleBackedProduct(base.BaseProduct):
    """TEST-ONLY plugin (lives in tests/): lets the runner protocol be exercised on a
    machine without a GPU.  Not part of the product."""

    calls = []

    def __init__(self, *, kernel, dimension, normalize_rows=False, precision="float64"):
        super().__init__(kernel=kernel, dimension=dimension, normalize_rows=normalize_rows, precision=precision)
        self.name = f"OracleBackedProduct({precision})"
        OracleBackedProduct.calls.append("init")

    def prepare_data(self, *, source_points, target_points, same_points=False, density_estimation=False):
        self.y, self.x, self.de = source_points, (None if same_points else target_points), density_estimation
        OracleBackedProduct.calls.append("prepare_data")

    def fit(self):
        OracleBackedProduct.calls.append("fit")

    def prepare_query(self, *, source_signal):
        self.b = source_signal
        OracleBackedProduct.calls.append("prepare_query")

    def query(self):
        self.res = kmvp_oracle.product(kernel=self.kernel, source_points=self.y, target_points=self.x,
                                       source_signal=self.b, normalize_rows=self.normalize_rows,
                                       density_estimation=self.de, precision=self.precision)
        OracleBackedProduct.calls.append("query")

    def done(self):
        OracleBackedProduct.calls.append("done")


def test_runner_protocol_and_result_files(tmp_path):
    name = "product-cube-D3-E1-M200-N200-gaussian"
    y, b = datasets.cube_points(200, 3)
    truth = kmvp_oracle.product(kernel="gaussian", source_points=y, source_signal=b)
    data_root = str(tmp_path / "data")
    datasets.write_dataset(filename=datasets.dataset_path(name, data_root), task="product", kernel="gaussian",
                           source_points=y, source_signal=b, target_signal=truth)
    yaml_file = tmp_path / "algos.yaml"
    yaml_file.write_text(
        "oracle-backed:\n  hardware: CPU\n  product: true\n  docker-tag: none\n"
        "  module: test_abi_and_host\n  constructor: OracleBackedProduct\n  run-groups:\n    g:\n"
        "      datasets: ['*-gaussian']\n      args: [{precision: float64}, {precision: float32}]\n")
    OracleBackedProduct.calls.clear()
    out = runner.run_dataset(name, hardware="CPU", runs=2, definition_file=str(yaml_file),
                             data_root=data_root, results_root=str(tmp_path / "results"), verbose=False)
    assert len(out) == 2
    # runner.py:70-148: fresh instance per fit run, prepare_query before every query run
    one = ["init", "prepare_data", "fit"] * 2
    assert OracleBackedProduct.calls[:6] == one
    assert OracleBackedProduct.calls.count("query") == 4 and OracleBackedProduct.calls.count("prepare_query") == 4
    assert OracleBackedProduct.calls.count("done") >= 2
    loaded = list(results.load_all_results(name, root=str(tmp_path / "results")))
    assert len(loaded) == 2
    for (fn, attrs, result) in out:
        assert os.path.exists(fn)
        for key in ("dataset", "algo", "name", "kernel", "run_count", "build_time", "query_time", "memory_footprint"):
            assert key in attrs  # runner.py:151-163
        assert attrs["algo"] == "oracle-backed" and attrs["run_count"] == 2
    f = storage.open_file(out[0][0], "r")
    try:
        err = np.asarray(f["error"][:])
        assert metrics.result_errors(err)["max"] < 1e-12  # fp64 run reproduces the dataset
    finally:
        f.close()
    pts, front = metrics.summarize_results(name, root=str(tmp_path / "results"))
    assert len(pts) == 2 and len(front["front"]["labels"]) >= 1


def test_config1_plumbing_at_1e4_on_cpu(tmp_path):
    """BASELINE config 1 (Gaussian, uniform-3D, N = M = 1e4, E = 1, float64, CPU plumbing): the dataset recipe ->
    registry -> runner protocol -> result file chain at the config's own size, with the oracle-backed TEST plugin
    standing where the GPU plugin stands (no GPU here).  The truth comes from the C oracle, the run from the numpy
    oracle: two restatements of bruteforce.py:25-58,130-153 that must agree to rounding at 1e8 pairs."""
    import c_oracle

    n = 10000
    name = f"product-cube-D3-E1-M{n}-N{n}-gaussian"
    y, b = datasets.cube_points(n, 3)  # seed 10003 (datasets.py:258)
    yo, bo = kmvp_oracle.uniform_cube(n, 3)
    assert np.array_equal(y, yo) and np.array_equal(b, bo)
    truth = c_oracle.product(kernel="gaussian", source_points=y, source_signal=b, rows=np.arange(n))
    data_root = str(tmp_path / "data")
    datasets.write_dataset(filename=datasets.dataset_path(name, data_root), task="product", kernel="gaussian",
                           source_points=y, source_signal=b, target_signal=truth)
    yaml_file = tmp_path / "algos.yaml"
    yaml_file.write_text(
        "oracle-backed:\n  hardware: CPU\n  product: true\n  docker-tag: none\n"
        "  module: test_abi_and_host\n  constructor: OracleBackedProduct\n  run-groups:\n    g:\n"
        "      datasets: ['*-gaussian']\n      args: [{precision: float64}]\n")
    out = runner.run_dataset(name, hardware="CPU", runs=1, definition_file=str(yaml_file), data_root=data_root,
                             results_root=str(tmp_path / "results"), verbose=False)
    (fn, attrs, result), = out
    assert result.shape == (n, 1) and attrs["run_count"] == 1 and attrs["dataset"] == name
    f = storage.open_file(fn, "r")
    try:
        err = metrics.result_errors(np.asarray(f["error"][:]))
    finally:
        f.close()
    assert err["max"] / np.max(np.abs(truth)) < 1e-12, err


def test_bench_roofline_tables_and_committed_profiles():
    """bench.py's per-kernel roofline table names every pair-loop kernel the library can report for the BASELINE
    configs, and every config's traffic figure points at a committed rocprofv3 summary of THAT kernel."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)  # no GPU work at import
    blob = open(_lib.LIB_PATH, "rb").read()
    for kname, (bound, flops_per_pair, peak, basis) in bench.ROOF.items():
        assert kname.encode() in blob, f"{kname} is not a kernel of libkmvp.so"
        assert bound in ("mfma", "valu", "hbm") and flops_per_pair > 0 and peak > 0 and len(basis) > 20
    assert bench.ROOF["cellmm_kernel"][0] == "mfma" and bench.ROOF["cellmm_kernel"][2] == bench.PEAK_F16_MFMA_TFLOPS == 2500.0
    for kname, tag in (("cellmm_kernel", "gaussian_1e6_f32"), ("mfma_pipe_kernel", "c3_absexp_bf16"),
                       ("cfast_kernel", "c4shard_invdist_f32"), ("cell64_kernel", "c5_gaussian_1e5_f64")):
        traffic, source = bench.traffic_from_profile(kname, tag)
        assert traffic and traffic > 1e6, (kname, tag)
        assert os.path.exists(os.path.join(ROOT, source["file"])) and source["measured_in_this_run"] is False
        md = os.path.join(ROOT, source["file"].replace("_traffic.json", ".md"))
        # the summary is of the same kernel (cellmm_kernel names both MFMA shapes: cellmm_kernel<TT>, cellmm16_kernel<TT>)
        text = open(md).read()
        assert kname in text or (kname == "cellmm_kernel" and "cellmm16_kernel" in text), md
    assert bench.traffic_from_profile("no_such_kernel", "gaussian_1e6_f32") == (None, None)


def test_tools_and_entry_points_parse():
    """Every script under tools/ (probes, sweeps, profile summarisers), bench.py and __graft_entry__.py is valid Python and has
    a docstring saying what it measures: they are run by hand on a GPU box, nothing else keeps them alive."""
    import ast
    import glob

    paths = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    assert len(paths) > 20
    for p in paths:
        tree = ast.parse(open(p).read(), filename=p)
        assert ast.get_docstring(tree), f"{os.path.relpath(p, ROOT)} has no docstring"
