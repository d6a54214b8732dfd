"""CPU tests: the oracle (numpy and C restatements) against the reference's outputs."""
import numpy as np
import pytest

import c_oracle
import golden_cases
import kmvp_oracle
from conftest import rel_err

CASES = golden_cases.product_cases()
IDS = [c["name"] for c in CASES]


def _same_finite_pattern(a, b):
    return np.array_equal(np.isfinite(a), np.isfinite(b))


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_numpy_oracle_matches_reference(case, expected):
    y, x, b = golden_cases.make_inputs(case)
    for tag, precision, fast, tol in (
        ("f64", np.float64, False, 1e-12), ("f64fast", np.float64, True, 1e-6),
        ("f32", np.float32, False, 2e-4), ("f32fast", np.float32, True, 5e-2),
    ):
        want = expected[f"{case['name']}/{tag}"].astype(np.float64)
        with np.errstate(all="ignore"):
            got = kmvp_oracle.product(
                kernel=case["kernel"], source_points=y, target_points=x, source_signal=b,
                normalize_rows=case["normalize_rows"], density_estimation=case["density_estimation"],
                precision=precision, fast_sqdists=fast, block_rows=41)
        assert got.shape == want.shape and got.dtype == np.float64
        assert _same_finite_pattern(got, want), (case["name"], tag)
        assert rel_err(got, want) <= tol, (case["name"], tag, rel_err(got, want))


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_c_oracle_matches_reference(case, expected):
    y, x, b = golden_cases.make_inputs(case)
    want = expected[f"{case['name']}/f64"]
    got = c_oracle.product(
        kernel=case["kernel"], source_points=y, target_points=x, source_signal=b,
        normalize_rows=case["normalize_rows"], density_estimation=case["density_estimation"])
    assert got.shape == want.shape
    assert _same_finite_pattern(got, want), case["name"]
    assert rel_err(got, want) <= 1e-12, (case["name"], rel_err(got, want))
    want32 = expected[f"{case['name']}/f32"].astype(np.float64)
    got32 = c_oracle.product(
        kernel=case["kernel"], source_points=y, target_points=x, source_signal=b,
        normalize_rows=case["normalize_rows"], density_estimation=case["density_estimation"],
        precision=np.float32)
    assert rel_err(got32, want32) <= 2e-4, (case["name"], rel_err(got32, want32))


def test_row_subset_and_shards_agree_with_full():
    case = dict(N=257, M=193, D=3, E=3, seed=7, same_points=False, density_estimation=False)
    y, x, b = golden_cases.make_inputs(case)
    rows = np.array([0, 5, 192, 193, 194, 256])
    for kernel in golden_cases.KERNELS:
        full = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b)
        sub = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b, rows=rows)
        np.testing.assert_allclose(sub, full[rows], rtol=1e-13)
        csub = c_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b, rows=rows)
        np.testing.assert_allclose(csub, full[rows], rtol=1e-12)
        # two source shards, global zero pattern kept through j_offset / M_total
        num = np.zeros((257, 3))
        den = np.zeros((257, 1))
        for lo, hi in ((0, 100), (100, 193)):
            n_, d_ = kmvp_oracle.product(kernel=kernel, source_points=y[lo:hi], target_points=x,
                                         source_signal=b[lo:hi], j_offset=lo, M_total=193, raw_sums=True)
            num += n_
            den += d_
        np.testing.assert_allclose(num, full, rtol=1e-12)
        normed = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x,
                                     source_signal=b, normalize_rows=True)
        np.testing.assert_allclose(num / den, normed, rtol=1e-12)


def test_zero_column_rule_is_the_flat_index_rule():
    # bruteforce.py:13-14: buffer[::M+1] = 0 on the row-major (N, M) matrix
    for N, M in ((5, 3), (3, 5), (4, 4), (9, 2), (1, 1), (7, 1)):
        k = np.ones((N, M))
        k.reshape(-1)[:: M + 1] = 0
        mine = np.ones((N, M))
        jz = kmvp_oracle.zero_column(np.arange(N), M)
        for i in range(N):
            if jz[i] >= 0:
                mine[i, jz[i]] = 0
        assert np.array_equal(k, mine), (N, M)


def test_known_answer_identities():
    y, b = kmvp_oracle.uniform_cube(50, 3)
    for kernel in golden_cases.KERNELS:
        ones = kmvp_oracle.product(kernel=kernel, source_points=y, normalize_rows=True,
                                   density_estimation=True)
        assert np.array_equal(ones, np.ones((50, 1)))  # bruteforce.py:134-138
        K = kmvp_oracle.kernel_matrix(kernel=kernel, source_points=y)
        assert np.allclose(K, K.T)  # symmetric when x == y
        a = kmvp_oracle.product(kernel=kernel, source_points=y, source_signal=b, normalize_rows=True)
        assert a.min() >= b.min() - 1e-12 and a.max() <= b.max() + 1e-12  # convex combination
    one = kmvp_oracle.product(kernel="gaussian", source_points=np.zeros((1, 3)), density_estimation=True)
    assert one[0, 0] == 1.0


def test_uniform_cube_recipe_is_the_references():
    # datasets.py:258-266 uses the global numpy.random.seed(n + D) stream
    n, D = 100, 3
    np.random.seed(n + D)
    y_ref = 1 * np.random.rand(n, D)
    b_ref = np.random.randn(n, 1)
    y, b = kmvp_oracle.uniform_cube(n, D)
    assert np.array_equal(y, y_ref) and np.array_equal(b, b_ref)


def test_solver_oracle(expected):
    for case in golden_cases.solver_cases():
        y, _ = golden_cases.make_solver_inputs(case)
        a = expected[f"{case['name']}/a"]
        sol = kmvp_oracle.solve(kernel=case["kernel"], source_points=y, target_signal=a)
        res = kmvp_oracle.relative_residual(kernel=case["kernel"], source_points=y, solution=sol,
                                            target_signal=a)
        assert res < 1e-9
        ref = expected[f"{case['name']}/b_lstsq"]
        res_ref = kmvp_oracle.relative_residual(kernel=case["kernel"], source_points=y, solution=ref,
                                                target_signal=a)
        assert res_ref < 1e-9


def test_result_errors_definition():
    err = np.array([[3.0, 4.0], [0.0, 0.0], [1.0, 0.0]])
    m = kmvp_oracle.result_errors(err)  # metrics.py:53-59
    assert m["max"] == 5.0 and m["median"] == 1.0
    assert abs(m["mean"] - 2.0) < 1e-15 and abs(m["rmse"] - np.sqrt(26 / 3)) < 1e-15
