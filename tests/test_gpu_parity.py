"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the
plugin classes and the C ABI, against the reference's outputs (golden fixtures), the
oracle, and size-independent properties at the benchmark's full size.

Tolerances (stated once, used everywhere):
  float64  rel <= 1e-11  (same arithmetic as the reference's truth, other summation order)
  float32  rel <= max(1e-5, 2 x the error of the REFERENCE's own float32 run)
           -- 1e-5 is the north-star tolerance; rel = max_i ||err_i|| / max_i ||a_i||
           (metrics.py:53-56 norm, made relative).
Rows that are non-finite in the reference (coincident off-diagonal points under
inverse-distance, bruteforce.py:8-15) must be non-finite here too.
"""
import numpy as np
import pytest

import c_oracle
import golden_cases
import kmvp_oracle
from conftest import rel_err
from kernel_matrix_benchmarks_amd import _lib
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct, MI355XSolver

pytestmark = pytest.mark.gpu

CASES = golden_cases.product_cases()
IDS = [c["name"] for c in CASES]
TOL64 = 1e-11
TOL32 = 1e-5


def run_plugin(case, y, x, b, precision, **options):
    """The runner's call order, runner.py:70-148."""
    algo = MI355XProduct(kernel=case["kernel"], dimension=case["D"],
                         normalize_rows=case.get("normalize_rows", False), precision=precision, **options)
    try:
        algo.prepare_data(source_points=y, target_points=(y if x is None else x),
                          same_points=np.bool_(x is None),
                          density_estimation=np.bool_(b is None))
        algo.fit()
        algo.prepare_query(source_signal=(np.ones((len(y), 1)) if b is None else b))
        algo.query()
        out = algo.get_result()
        extra = algo.get_additional()
        assert algo.get_memory_usage() > 0
    finally:
        algo.done()
    assert out.dtype == np.float64 and out.flags["C_CONTIGUOUS"]
    return out, extra


def row_finite(a):
    return np.isfinite(a).all(axis=-1)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_float64_matches_reference(case, expected):
    y, x, b = golden_cases.make_inputs(case)
    want = expected[f"{case['name']}/f64"]
    got, _ = run_plugin(case, y, x, b, np.float64)
    assert got.shape == want.shape
    assert np.array_equal(row_finite(got), row_finite(want)), "non-finite rows differ"
    assert rel_err(got, want) <= TOL64, rel_err(got, want)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_float32_matches_reference(case, expected):
    y, x, b = golden_cases.make_inputs(case)
    truth = expected[f"{case['name']}/f64"]
    ref32 = expected[f"{case['name']}/f32"].astype(np.float64)
    got, extra = run_plugin(case, y, x, b, "float32")  # precision arrives as a string, algos.yaml:157
    assert got.shape == truth.shape
    assert np.array_equal(row_finite(got), row_finite(truth)), "non-finite rows differ"
    tol = max(TOL32, 2 * rel_err(ref32, truth))
    assert rel_err(got, truth) <= tol, (rel_err(got, truth), tol)
    assert extra["n_gpus"] == 1 and extra["device_kernel"]


@pytest.fixture(scope="module")
def expected_f16():
    """Outputs of the reference's own float16 runs (tests/make_golden_f16.py)."""
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "expected_f16.npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_float16_is_at_least_as_accurate_as_the_references_float16(case, expected, expected_f16):
    """precision="float16" (algos.yaml:157,160): the reference casts points and signal to float16 and lets numpy
    do float16 arithmetic; the plugin rounds the inputs the same way and computes in float32.  Pinned two ways:
    (1) against the float64 oracle on the SAME float16-rounded inputs, at the float32 tolerance; (2) against the
    reference's own float16 output, by the rule used for float32 everywhere in this file: the plugin's distance from
    the float64 truth is at most 2 x the distance of the REFERENCE's own run in that precision (both carry the
    rounding of the inputs to float16, ~1e-3; the reference adds float16 arithmetic, which sometimes cancels part)."""
    y, x, b = golden_cases.make_inputs(case)
    r16 = lambda a: None if a is None else np.asarray(a, dtype=np.float16).astype(np.float64)
    truth16 = kmvp_oracle.product(kernel=case["kernel"], source_points=r16(y), target_points=r16(x), source_signal=r16(b),
                                  normalize_rows=case["normalize_rows"], density_estimation=case["density_estimation"])
    got, extra = run_plugin(case, y, x, b, "float16")
    assert got.shape == truth16.shape
    assert np.array_equal(row_finite(got), row_finite(truth16)), "non-finite rows differ (coincident float16 points)"
    assert rel_err(got, truth16) <= TOL32, rel_err(got, truth16)
    truth = expected[f"{case['name']}/f64"]
    ref16 = expected_f16[f"{case['name']}/f16"].astype(np.float64)
    both = row_finite(truth) & row_finite(ref16) & row_finite(got)
    if both.any():
        mine, theirs = rel_err(got[both], truth[both]), rel_err(ref16[both], truth[both])
        assert mine <= 2 * theirs + TOL32, (mine, theirs)


def test_exp_dot_attention_matches_direct_evaluation():
    """k(x, y) = exp(<x, y>) (README.md:51-59; PARITY UNPINNED: no reference plugin implements it) against a direct
    float64 evaluation: softmax attention (row-normalised, E value channels), plain products, densities; float32 / float16
    at D <= 64 on the native online-max kernel (fastmm_kernel, ONLINE = 1), bfloat16 on mfma_pipe_kernel's, float64 through the Gaussian identity;
    targets != sources; key norms spanning |y|^2/2 up to ~60."""
    rs = np.random.RandomState(99)
    native = ("fastmm_kernel",)
    shapes = [(3, 40000, 40000, 4, "float32", 1.0, native), (16, 700, 900, 8, "float32", 1.0, native),
              (16, 700, 900, 8, np.float64, 2.5, None), (64, 1024, 2048, 64, "float32", 0.35, native),
              (64, 1024, 2048, 64, "bfloat16", 0.35, ("mfma_pipe_kernel",)), (3, 500, 300, 1, "float16", 1.0, native)]
    for D, N, M, E, precision, spread, want_kernel in shapes:
        y = rs.randn(M, D) * spread / np.sqrt(D) * (1.5 if D == 3 else 3.0)
        x = rs.randn(N, D) * spread / np.sqrt(D) * (1.5 if D == 3 else 3.0)
        if D == 3:
            y, x = rs.rand(M, D), rs.rand(N, D)  # a dense cloud inside the radius rule: the matrix-core forms take it
        b = rs.randn(M, E)
        tol = {"bfloat16": TOL_BF16, "float16": 5e-3}.get(precision, TOL64 * 10 if precision is np.float64 else 2 * TOL32)
        rows = rs.choice(N, size=min(N, 300), replace=False)
        for norm in (True, False):
            algo = MI355XProduct(kernel="exp-dot", dimension=D, normalize_rows=norm, precision=precision)
            try:
                algo.prepare_data(source_points=y, target_points=x, same_points=False)
                algo.fit()
                algo.prepare_query(source_signal=b)
                algo.query()
                got = algo.get_result()
                kname = algo.device_kernel
            finally:
                algo.done()
            want = kmvp_oracle.exp_dot_product(source_points=y, target_points=x[rows], source_signal=b, normalize_rows=norm)
            assert got.shape == (N, E) and got.dtype == np.float64
            assert rel_err(got[rows], want) <= tol, (D, precision, norm, kname, rel_err(got[rows], want))
            if want_kernel:
                assert kname in want_kernel, kname
            if norm:  # softmax rows are convex combinations of the value rows
                assert got.min() >= b.min() - 1e-3 and got.max() <= b.max() + 1e-3
    # density (b = 1) and the all-ones shortcut of normalised densities; same_points
    y = rs.randn(400, 5) * 0.6
    for norm in (False, True):
        algo = MI355XProduct(kernel="exp-dot", dimension=5, normalize_rows=norm, precision=np.float64)
        try:
            algo.prepare_data(source_points=y, target_points=y, same_points=True, density_estimation=True)
            algo.prepare_query(source_signal=np.ones((400, 1)))
            algo.query()
            got = algo.get_result()
        finally:
            algo.done()
        want = kmvp_oracle.exp_dot_product(source_points=y, normalize_rows=norm)
        assert rel_err(got, want) <= 1e-10
    # the solver: K b = a with K = exp(<x_i, x_j>) on a small well-separated cloud
    ys = rs.rand(200, 3) * 5.0
    b_true = rs.randn(200, 1)
    a = kmvp_oracle.exp_dot_product(source_points=ys, source_signal=b_true)
    sol = MI355XSolver(kernel="exp-dot", dimension=3, precision=np.float64, rtol=1e-8, maxit=20000)
    try:
        sol.prepare_data(source_points=ys)
        sol.fit()
        sol.prepare_query(target_signal=a)
        sol.query()
        bs = sol.get_result()
        info = sol.get_additional()
    finally:
        sol.done()
    back = kmvp_oracle.exp_dot_product(source_points=ys, source_signal=bs)
    true_res = np.linalg.norm(back - a) / np.linalg.norm(a)
    assert true_res <= 1e-6, (true_res, info)
    # the verdict is taken on the UNSCALED system K b = a (one more product), not on the scaled one CG iterated on
    assert abs(info["cg_relative_residual"] - true_res) <= 0.5 * true_res + 1e-12, (true_res, info)
    assert info["cg_converged"] == bool(info["cg_relative_residual"] <= 1.5e-8) and "cg_scaled_system_residual" in info


def test_exp_dot_float64_products_whose_target_factor_alone_overflows():
    """The identity route multiplies the Gaussian product by exp(|x|^2/2 + c): that factor may leave float64 where
    exp(<x, y>) itself does not (found by tools/fuzz_parity.py, seed 77 case 1041).  Those rows are assembled in the log
    domain; the results must be finite and match the direct evaluation."""
    rs = np.random.RandomState(1041)
    D, N, M, E = 40, 300, 33, 5
    y, x, b = rs.rand(M, D) * 6.6, rs.rand(N, D) * 6.6, rs.randn(M, E)
    assert 0.5 * (x * x).sum(axis=1).max() + 0.5 * (y * y).sum(axis=1).max() > 780 > 700 > (x @ y.T).max()
    algo = MI355XProduct(kernel="exp-dot", dimension=D, precision=np.float64)
    try:
        algo.prepare_data(source_points=y, target_points=x, same_points=False)
        algo.fit()
        algo.prepare_query(source_signal=b)
        algo.query()
        got = algo.get_result()
    finally:
        algo.done()
    want = kmvp_oracle.exp_dot_product(source_points=y, target_points=x, source_signal=b)
    mass = kmvp_oracle.exp_dot_product(source_points=y, target_points=x, source_signal=np.abs(b))
    assert np.isfinite(want).all() and np.isfinite(got).all()
    assert float((np.abs(got - want) / mass).max()) <= 1e-9  # logits ~700 carry 700 x 1e-16 x a few roundings


def test_exp_dot_native_kernel_has_no_range_limit():
    """The online-max formulation (include/kmvp.h kmvp_expdot[_norm]): key norms |y|^2/2 up to ~500
    and logits <x, y> of several hundred either sign, ragged N != M, targets != sources.  Softmax rows against the direct
    float64 evaluation of the float32-rounded inputs; plain products row by row in RELATIVE terms (the rows span hundreds
    of orders of magnitude); logits beyond float64's exp range: the plain product is inf exactly where numpy's is, the
    softmax stays finite.  The Gaussian identity refuses the same inputs (range check) instead of dropping sources."""
    rs = np.random.RandomState(2024)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    for D, N, M, E, sy, sx in ((8, 777, 1333, 5, 3.5, 1.0), (24, 300, 4099, 3, 6.0, 0.6), (64, 130, 1000, 33, 3.9, 0.5), (2, 50, 70, 1, 22.0, 1.0)):
        y, x, b = f32(rs.randn(M, D) * sy), f32(rs.randn(N, D) * sx), f32(rs.randn(M, E))
        assert 150 < np.max(np.sum(y * y, axis=1)) / 2, "the case is meant to have large key norms"
        logits = x @ y.T
        for norm in (True, False):
            algo = MI355XProduct(kernel="exp-dot", dimension=D, normalize_rows=norm, precision="float32")
            try:
                algo.prepare_data(source_points=y, target_points=x, same_points=False)
                algo.fit()
                algo.prepare_query(source_signal=b)
                algo.query()
                got = algo.get_result()
                assert algo.device_kernel == "fastmm_kernel" and "online shift" in algo.get_additional()["dispatch_note"]
            finally:
                algo.done()
            want = kmvp_oracle.exp_dot_product(source_points=y, target_points=x, source_signal=b, normalize_rows=norm)
            # float32 logits carry an absolute error ~ eps32 sum_d |x_d y_d|: the weights are good to that, relatively
            lerr = float(np.max(np.abs(x) @ np.abs(y).T)) * 2.0 ** -23
            tol = max(2 * TOL32, 8 * lerr)
            if norm:
                assert np.isfinite(got).all()
                assert rel_err(got, want) <= tol, (D, norm, rel_err(got, want), tol)
            else:
                fin = np.isfinite(want).all(axis=1)
                assert np.array_equal(np.isfinite(got).all(axis=1), fin), "plain product: inf rows differ from numpy's"
                # a row's yardstick is sum_j k |b_j|: two large terms of opposite sign may cancel in the row itself
                mass = kmvp_oracle.exp_dot_product(source_points=y, target_points=x, source_signal=np.abs(b))
                rowscale = np.max(mass[fin], axis=1, keepdims=True)
                assert np.max(np.abs(got[fin] - want[fin]) / rowscale) <= tol, (D, norm, tol)
        print(f"exp-dot native D={D}: logits in [{logits.min():.0f}, {logits.max():.0f}], |y|^2/2 up to {np.max(np.sum(y*y,1))/2:.0f}")
    # beyond float64: logits ~ 2000.  softmax finite and right, plain product inf like numpy
    y, x, b = f32(rs.randn(500, 4) * 30.0), f32(rs.randn(64, 4) * 30.0), f32(rs.randn(500, 2))
    for norm in (True, False):
        algo = MI355XProduct(kernel="exp-dot", dimension=4, normalize_rows=norm, precision="float32")
        try:
            algo.prepare_data(source_points=y, target_points=x, same_points=False)
            algo.prepare_query(source_signal=b)
            algo.query()
            got = algo.get_result()
        finally:
            algo.done()
        with np.errstate(over="ignore", invalid="ignore"):
            want = kmvp_oracle.exp_dot_product(source_points=y, target_points=x, source_signal=b, normalize_rows=norm)
        if norm:
            assert np.isfinite(got).all() and rel_err(got, want) <= 1e-2, rel_err(got, want)  # near one-hot rows; logit error ~1e-3
        else:
            big = np.max(x @ y.T, axis=1) > 720
            assert big.any() and not np.isfinite(got[big]).any()
    # the identity route says so instead of zeroing small-norm sources
    with pytest.raises(NotImplementedError, match="spans"):
        algo = MI355XProduct(kernel="exp-dot", dimension=70, normalize_rows=True, precision="float32")  # D > 64: identity
        try:
            algo.prepare_data(source_points=rs.randn(100, 70) * 2.0, target_points=rs.randn(10, 70), same_points=False)
        finally:
            algo.done()


def bf16_round(a, c=None):
    """What the bf16 packing kernels make of the plugin's float32 inputs, as float64: (float32 a) x (float32 c) -- ONE float32
    product, as the kernel forms it -- rounded to the nearest bfloat16 (ties to even), then divided by c again.  (Multiplying in
    float64 instead lands on the other side of a bf16 rounding boundary for about one operand in a million, and a logit of
    ~1000 then moves by half a unit: a one-row artefact of the emulation that looked like a kernel defect.)"""
    v = np.ascontiguousarray(a, dtype=np.float32)
    if c is not None:
        v = v * np.float32(c)
    u = np.ascontiguousarray(v, dtype=np.float32).view(np.uint32)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    r = u.view(np.float32).astype(np.float64)
    return r if c is None else r / c


def test_exp_dot_bfloat16_native_kernel_with_online_shift():
    """bfloat16 exp(<x, y>) on mfma_pipe_kernel / mfma_kernel with the per-target running shift (kmvp_mfma.hpp): softmax
    attention and plain products against the direct float64 evaluation (PARITY UNPINNED, as every exp-dot check), on logits
    far outside what the Gaussian identity could take (|y|^2/2 spread in the hundreds, logits from -2000 to +2000), with
    the largest logits arriving LATE in the source order (the shift has to move while sums are under way), several source
    segments, ragged sizes, E beyond one column tile, and a signal of ones.  The yardstick is the bf16 tolerance of this
    file on inputs pre-rounded to what the kernel sees (the operands are x, y times sqrt(log2 e) rounded to bf16: a logit
    carries ~2^-9 |x||y| / sqrt(D) of that rounding, so wide-logit rows are compared with their logits' own uncertainty)."""
    rs = np.random.RandomState(515)
    seen = set()
    for D, N, M, E, scale, late in ((64, 1000, 4096, 64, 0.35, False), (64, 777, 5000, 64, 1.0, True), (16, 300, 2050, 8, 12.0, True),
                                    (100, 200, 1500, 96, 0.6, False), (8, 64, 33, 1, 6.0, True), (64, 4096, 65536, 16, 0.5, True)):
        y = rs.randn(M, D) * scale
        x = rs.randn(N, D) * scale
        if late:  # sources sorted by norm: the big logits come last
            y = y[np.argsort((y * y).sum(axis=1))]
        b = rs.randn(M, E)
        for norm in (True, False):
            algo = MI355XProduct(kernel="exp-dot", dimension=D, normalize_rows=norm, precision="bfloat16")
            try:
                algo.prepare_data(source_points=y, target_points=x, same_points=False)
                algo.fit()
                algo.prepare_query(source_signal=b)
                algo.query()
                got = algo.get_result()
                seen.add(algo.device_kernel)
                note = algo.get_additional().get("dispatch_note", "")
            finally:
                algo.done()
            assert "online shift" in note, note
            rows = rs.choice(N, size=min(N, 200), replace=False)
            # the truth on the operands the kernel multiplies: (x c) and (y c) rounded to bf16, c = sqrt(log2 e)
            c = 1.2011224087864498
            xr, yr = bf16_round(x[rows], c), bf16_round(y, c)
            with np.errstate(over="ignore", invalid="ignore"):
                want = kmvp_oracle.exp_dot_product(source_points=yr, target_points=xr, source_signal=b, normalize_rows=norm)
                mass = want if norm else kmvp_oracle.exp_dot_product(source_points=yr, target_points=xr, source_signal=np.abs(b))
            live = np.isfinite(want).all(axis=1) & np.isfinite(mass).all(axis=1)
            if norm:
                assert live.all(), (D, scale)  # softmax rows exist whatever the logits
            elif not live.any():
                continue  # every plain product of this shape leaves float64
            assert np.isfinite(got[rows][live]).all(), (D, N, M, E, norm)
            yard = np.abs(want[live]).max() if norm else np.abs(mass[live])
            err = float((np.abs(got[rows][live] - want[live]) / yard).max())
            assert err <= TOL_BF16, (D, N, M, E, scale, norm, err)
            if norm:
                assert got.min() >= b.min() - 1e-2 and got.max() <= b.max() + 1e-2
    assert seen == {"mfma_pipe_kernel", "mfma_kernel"}, seen


def test_bfloat16_gaussian_targets_far_from_every_source():
    """The bf16 Gaussian with targets != sources carries exp(<x,y>)'s per-target running shift (K_GAUSSIAN_SHIFTED in
    kmvp_mfma.hpp): targets 3 ... 25 kernel lengths away from every source (kernel values down to 2^-900: the whole row far
    under the float32 range) keep bf16 accuracy row by row -- normalised rows stay convex combinations instead of 0/0, plain
    products come back at their own scale in float64.  Also sources sorted so that the nearest ones arrive LAST (the shift has
    to move while sums are under way), both kernels (mfma_pipe_kernel / mfma_kernel), several segments; targets == sources keep
    the plain kernel.  Truth: the float64 oracle on the operands the kernel multiplies (points x sqrt(log2 e) rounded to bf16)."""
    rs = np.random.RandomState(717)
    c = 1.2011224087864498
    seen = set()
    for D, N, M, E, offset in ((16, 500, 3000, 8, 3.0), (64, 300, 4096, 64, 9.0), (64, 257, 2000, 33, 25.0), (100, 100, 1500, 96, 6.0),
                               (24, 4096, 40000, 4, 12.0)):
        y = rs.rand(M, D) / np.sqrt(D / 3.0)
        u = rs.randn(D)
        x = rs.rand(N, D) / np.sqrt(D / 3.0) + offset * u / np.linalg.norm(u)
        # nearest sources last: sort the sources by their distance to the targets' centre, descending
        y = y[np.argsort(-((y - x.mean(axis=0)) ** 2).sum(axis=1))]
        b = rs.randn(M, E)
        yr, xr = bf16_round(y, c), bf16_round(x, c)
        for norm in (True, False):
            algo = MI355XProduct(kernel="gaussian", dimension=D, normalize_rows=norm, precision="bfloat16")
            try:
                algo.prepare_data(source_points=y, target_points=x, same_points=False)
                algo.fit()
                algo.prepare_query(source_signal=b)
                algo.query()
                got = algo.get_result()
                seen.add(algo.device_kernel)
                note = algo.get_additional().get("dispatch_note", "")
            finally:
                algo.done()
            assert "online shift" in note, note
            rows = rs.choice(N, size=min(N, 200), replace=False)
            want = kmvp_oracle.product(kernel="gaussian", source_points=yr, target_points=xr[rows], source_signal=b, normalize_rows=norm)
            mass = kmvp_oracle.product(kernel="gaussian", source_points=yr, target_points=xr[rows], source_signal=np.abs(b),
                                       normalize_rows=norm)
            assert np.isfinite(want).all() and (mass > 0).all()
            assert np.isfinite(got).all(), (D, offset, norm)
            err = float((np.abs(got[rows] - want) / mass).max())
            assert err <= TOL_BF16, (D, N, M, E, offset, norm, err)
            if norm:
                assert got.min() >= b.min() - 1e-2 and got.max() <= b.max() + 1e-2
    assert seen == {"mfma_pipe_kernel", "mfma_kernel"}, seen
    # targets == sources: the plain kernel (every row holds k = 1)
    y = rs.rand(2000, 64) / np.sqrt(64 / 3.0)
    algo = MI355XProduct(kernel="gaussian", dimension=64, normalize_rows=True, precision="bfloat16")
    try:
        algo.prepare_data(source_points=y, target_points=y, same_points=True)
        algo.prepare_query(source_signal=rs.randn(2000, 4))
        algo.query()
        assert "online shift" not in algo.get_additional().get("dispatch_note", "")
    finally:
        algo.done()


def test_exp_dot_bfloat16_edge_cases():
    """The bf16 exp(<x, y>) kernels at their edges: one source / one target, targets == sources, density estimation (the
    plugin passes a signal of ones), the widest instantiated shape (D = 141, E = 128: mfma_kernel), a D beyond it (refused,
    not mis-served), and non-finite coordinates: a NaN target poisons its own row only, every other row stays right."""
    rs = np.random.RandomState(616)
    c = 1.2011224087864498

    def run(y, x, b, norm, same=False, dens=False):
        algo = MI355XProduct(kernel="exp-dot", dimension=y.shape[1], normalize_rows=norm, precision="bfloat16")
        try:
            algo.prepare_data(source_points=y, target_points=y if same else x, same_points=same, density_estimation=dens)
            algo.fit()
            algo.prepare_query(source_signal=b)
            algo.query()
            return algo.get_result(), algo.device_kernel
        finally:
            algo.done()

    def truth(y, x, b, norm):
        yr, xr = bf16_round(y, c), bf16_round(x, c)
        want = kmvp_oracle.exp_dot_product(source_points=yr, target_points=xr, source_signal=b, normalize_rows=norm)
        mass = want if norm else kmvp_oracle.exp_dot_product(source_points=yr, target_points=xr,
                                                              source_signal=None if b is None else np.abs(b))
        return want, mass

    for D, N, M, E in ((8, 1, 1, 1), (8, 70, 1, 3), (8, 1, 70, 3), (5, 33, 257, 2), (141, 100, 300, 128)):
        y, x, b = rs.randn(M, D) * 0.5, rs.randn(N, D) * 0.5, rs.randn(M, E)
        for norm in (True, False):
            got, kname = run(y, x, b, norm)
            want, mass = truth(y, x, b, norm)
            yard = np.abs(want).max() if norm else np.abs(mass)
            assert got.shape == (N, E) and float((np.abs(got - want) / yard).max()) <= TOL_BF16, (D, N, M, E, norm, kname)
    # targets == sources; density (a = sum_j k: the reference's density_estimation branch, bruteforce.py:146-149)
    y = rs.randn(500, 16) * 0.7
    got, _ = run(y, None, rs.randn(500, 2), True, same=True)
    assert np.isfinite(got).all()
    got, _ = run(y, None, None, False, same=True, dens=True)
    want, _ = truth(y, y, None, False)
    assert got.shape == (500, 1) and float((np.abs(got - want) / np.abs(want)).max()) <= TOL_BF16
    # beyond the widest instantiation: an error, never another kernel's answer
    with pytest.raises((NotImplementedError, RuntimeError)):
        run(rs.randn(50, 150), rs.randn(20, 150), rs.randn(50, 1), True)
    # non-finite coordinates
    y, x, b = rs.randn(300, 8), rs.randn(64, 8), rs.randn(300, 2)
    x[5, 3] = np.nan
    for norm in (True, False):
        got, _ = run(y, x, b, norm)
        ok = np.ones(64, dtype=bool)
        ok[5] = False
        want, mass = truth(y, x[ok], b, norm)
        yard = np.abs(want).max() if norm else np.abs(mass)
        assert np.isnan(got[5]).all()
        assert float((np.abs(got[ok] - want) / yard).max()) <= TOL_BF16, norm


def test_targets_far_from_every_source_keep_float32_accuracy():
    """the matrix-core forms with several signal columns store a kernel value as 2^15 k in two f16
    pieces.  With ONE global shift a target 4-5 away from every source had its whole row at or below the f16 floor (5e-2
    off, NaN rows when normalised).  Targets != sources now run the per-target online shift: disjoint clouds, targets
    offset by 4.2 / 4.8 / 5.5 / 9 along x, Gaussian and exp(-r), plain and normalised, inside the radius rule
    (fastmm_kernel, ONLINE = 1), outside it and exp(-r) (cfastmm_kernel, ONLINE = 1), exp(-r) at D = 6 (fastmm_kernel);
    against the float64 oracle under the float32 rule of this file."""
    rs = np.random.RandomState(77)
    seen = set()
    for kernel, D, box in (("gaussian", 3, 1.0), ("gaussian", 3, 0.2), ("absolute-exponential", 3, 1.0), ("absolute-exponential", 6, 0.5),
                           ("gaussian", 2, 0.2)):
        M, E = 6000, 8
        y = rs.rand(M, D) * box
        # (exp(-80) = 1.8e-35 is still a normal float32; beyond ~87 the float32 kernels flush where numpy denormalises)
        offs = np.array([4.2, 4.8, 5.5, 9.0] if kernel == "gaussian" else [4.2, 20.0, 45.0, 80.0])
        x = rs.rand(400, D) * box
        x[:, 0] += np.repeat(offs, 100)
        if box < 1:  # stay inside the radius rule (squared half-diagonal <= 8 / log2(e)): the expanded forms take it
            x[:, 0] = rs.rand(400) * box + np.repeat([1.3, 1.6, 2.0, 2.4], 100)
        b = rs.randn(M, E)
        for norm in (False, True):
            want = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b, normalize_rows=norm)
            ref32 = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b, normalize_rows=norm,
                                        precision=np.float32)
            got, extra = run_plugin(dict(kernel=kernel, D=D, normalize_rows=norm), y, x, b, "float32")
            assert "online shift" in extra["dispatch_note"] or extra["device_kernel"] not in ("fastmm_kernel", "cfastmm_kernel"), extra
            seen.add(extra["device_kernel"])
            # row by row: every group of targets has its own scale (e^-17 ... e^-80); a row's yardstick is sum_j k |b_j|
            mass = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=np.abs(b))
            if norm:
                mass = mass / kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=np.ones((M, 1)))
            fin = np.isfinite(want).all(axis=1) & (np.max(mass, axis=1) > 0)
            assert np.isfinite(got[fin]).all(), (kernel, D, box, norm, extra["device_kernel"])
            scale = np.max(mass[fin], axis=1, keepdims=True)
            err = np.max(np.abs(got[fin] - want[fin]) / scale)
            ok32 = fin & np.isfinite(ref32).all(axis=1)
            err32 = np.max(np.abs(ref32[ok32] - want[ok32]) / np.max(mass[ok32], axis=1, keepdims=True)) if ok32.any() else 0.0
            assert err <= max(TOL32, 2 * err32), (kernel, D, box, norm, extra["device_kernel"], err, err32)
    assert {"fastmm_kernel", "cfastmm_kernel"} <= seen, seen


LOW_D_E1 = [c for c in CASES if c["D"] <= 39 and (c["E"] == 1 or c["density_estimation"])]


@pytest.mark.parametrize("case", LOW_D_E1, ids=[c["name"] for c in LOW_D_E1])
def test_fast_sqdists_matches_reference(case, expected):
    """fast_sqdists=True: expanded squared distances on the matrix cores (split-bf16 MFMA).
    Same tolerance as the difference form for the Gaussian; for exp(-r) and 1/r the error of
    the expansion near coincident points is inherent (the reference's own fast form has it
    too), so the bound there is twice the error of the REFERENCE's float32 fast run."""
    y, x, b = golden_cases.make_inputs(case)
    truth = expected[f"{case['name']}/f64"]
    if not np.isfinite(truth).all():
        pytest.skip("coincident points: the expansion has no exact zero distance")
    got, extra = run_plugin(case, y, x, b, "float32", fast_sqdists=True)
    if not (case["normalize_rows"] and case["density_estimation"]):
        # (exp(-r) at 5 <= D <= 64 inside the radius rule: the same expansion with the closest pairs recomputed exactly --
        # fastmm_kernel, held to the Gaussian's tolerance)
        assert extra["device_kernel"] in ("fast_kernel", "fastmm_kernel")
    if case["kernel"] == "gaussian" or extra.get("device_kernel") == "fastmm_kernel":
        tol = TOL32
    else:
        # not smooth at s = 0: the absolute error of the expanded s is amplified near coincident
        # points -- in the reference's fast form as well, which calibrates the bound
        ref32fast = expected[f"{case['name']}/f32fast"].astype(np.float64)
        fin = np.isfinite(ref32fast).all(axis=-1)
        floor = 1e-3 if case["kernel"] == "inverse-distance" else 1e-4
        tol = max(floor, 2 * rel_err(ref32fast[fin], truth[fin]))
    assert rel_err(got, truth) <= tol, (rel_err(got, truth), tol)


CENTRED = [c for c in CASES if c["D"] <= 4 and (c["E"] == 1 or c["density_estimation"])
           and (c["kernel"] != "inverse-distance" or c["same_points"])]


@pytest.mark.parametrize("case", CENTRED, ids=[c["name"] for c in CENTRED])
def test_centred_sqdists_matches_reference(case, expected):
    """fast_sqdists="centred": expansion around the centre of each group of Morton-sorted sources
    plus exact recomputation of the closest pairs -- held to the tolerance of the difference form
    for EVERY kernel, non-finite rows (coincident points) included."""
    y, x, b = golden_cases.make_inputs(case)
    truth = expected[f"{case['name']}/f64"]
    ref32 = expected[f"{case['name']}/f32"].astype(np.float64)
    got, extra = run_plugin(case, y, x, b, "float32", fast_sqdists="centred")
    if not (case["normalize_rows"] and case["density_estimation"]):
        assert extra["device_kernel"] == "cfast_kernel"
    assert np.array_equal(row_finite(got), row_finite(truth)), "non-finite rows differ"
    tol = max(TOL32, 2 * rel_err(ref32, truth))
    assert rel_err(got, truth) <= tol, (rel_err(got, truth), tol)


CELLS = [c for c in CASES if c["D"] <= 3 and c["kernel"] == "gaussian" and (c["E"] == 1 or c["density_estimation"])]


@pytest.mark.parametrize("case", CELLS, ids=[c["name"] for c in CELLS])
def test_cell_kernel_matches_reference(case, expected):
    """fast_sqdists="cells": the Gaussian's exp() range-reduced by grid cells, the polynomial remainder
    on the matrix cores (kmvp_cell.hpp) -- held to the tolerance of the difference form."""
    y, x, b = golden_cases.make_inputs(case)
    truth = expected[f"{case['name']}/f64"]
    ref32 = expected[f"{case['name']}/f32"].astype(np.float64)
    for form in ("cells", "cells-valu"):
        for tiles in (1, 2, 4, 8):
            got, extra = run_plugin(case, y, x, b, "float32", fast_sqdists=form, fast_tiles=tiles)
            if not (case["normalize_rows"] and case["density_estimation"]):
                # "cells": cellmm_kernel (sum over the sources in the MFMA accumulator; normalised rows = a second
                # launch with b = 1) on clouds inside the radius rule, cell_kernel otherwise; "cells-valu": cell_kernel
                allowed = ("cell_kernel",) if form == "cells-valu" else ("cellmm_kernel", "cell_kernel")
                assert extra["device_kernel"] in allowed, (form, extra)
                CELL_KERNELS_SEEN.add(extra["device_kernel"])
            tol = max(TOL32, 2 * rel_err(ref32, truth))
            assert rel_err(got, truth) <= tol, (form, tiles, rel_err(got, truth), tol)


CELL_KERNELS_SEEN = set()

CELLS_MULTI = [c for c in CASES if c["D"] <= 3 and c["kernel"] == "gaussian" and c["E"] > 1 and not c["density_estimation"]]


@pytest.mark.parametrize("case", CELLS_MULTI, ids=[c["name"] for c in CELLS_MULTI])
def test_cellmm_kernel_with_several_signal_columns_matches_reference(case, expected):
    """E > 1 (low-D attention with E value channels, bruteforce.py:142-145): one cellmm_kernel launch per signal
    column, plus one with b = 1 for the denominator of normalised rows."""
    y, x, b = golden_cases.make_inputs(case)
    truth = expected[f"{case['name']}/f64"]
    ref32 = expected[f"{case['name']}/f32"].astype(np.float64)
    for tiles in (1, 8):
        got, extra = run_plugin(case, y, x, b, "float32", fast_sqdists="cells", fast_tiles=tiles)
        assert extra["device_kernel"] in ("cellmm_kernel", "lowd_kernel"), extra  # (wide clouds: the difference form)
        CELL_KERNELS_SEEN.add("E>1:" + extra["device_kernel"])
        tol = max(TOL32, 2 * rel_err(ref32, truth))
        assert got.shape == truth.shape and rel_err(got, truth) <= tol, (tiles, rel_err(got, truth), tol)


FMM_MULTI = [c for c in CASES if c["D"] <= 64 and c["kernel"] == "gaussian" and c["E"] > 1 and not c["density_estimation"]]


@pytest.mark.parametrize("case", FMM_MULTI, ids=[c["name"] for c in FMM_MULTI])
def test_fastmm_kernel_with_several_signal_columns_matches_reference(case, expected):
    """E > 1, fast_sqdists=True (bruteforce.py:36-49 with an (M, E) signal, :142-153): both matrix products on the matrix
    cores (kmvp_fastmm.hpp) -- one pass for all the columns plus the denominator of normalised rows -- held to the
    tolerance of the difference form, for every tile count."""
    y, x, b = golden_cases.make_inputs(case)
    truth = expected[f"{case['name']}/f64"]
    ref32 = expected[f"{case['name']}/f32"].astype(np.float64)
    results = []
    for tiles in (1, 2, 4):
        got, extra = run_plugin(case, y, x, b, "float32", fast_sqdists=True, fast_tiles=tiles)
        assert extra["device_kernel"] == "fastmm_kernel", extra
        tol = max(TOL32, 2 * rel_err(ref32, truth))
        assert got.shape == truth.shape and rel_err(got, truth) <= tol, (tiles, rel_err(got, truth), tol)
        results.append(got)
    # the sums of a target do not depend on how many tiles its wavefront owns
    assert np.array_equal(results[0], results[1]) and np.array_equal(results[0], results[2])


@pytest.mark.parametrize("D,E,norm", [(3, 40, True), (2, 32, True), (8, 33, False), (1, 17, False), (7, 64, True), (16, 20, True), (39, 5, False),
                                         (3, 16, True), (5, 16, True), (3, 16, False), (64, 33, True), (50, 1, False), (40, 2, True)])
def test_fastmm_kernel_column_blocks_and_ragged_sizes(D, E, norm):
    """More than 32 columns run as blocks of 32 (the denominator is the last column of the last block); N and M are
    multiples of nothing; the columns' scales span twelve decades (each column is scaled by its own power of two before
    the f16 split, so that a small column is as accurate as a large one).  Checked against the float64 oracle."""
    rng = np.random.RandomState(100 * D + E)
    n, m = 1237, 2051
    y = rng.rand(m, D) / np.sqrt(max(D, 3) / 3.0)
    x = rng.rand(n, D) / np.sqrt(max(D, 3) / 3.0)
    b = rng.randn(m, E) * 10.0 ** rng.randint(-6, 7, size=E)
    algo = MI355XProduct(kernel="gaussian", dimension=D, normalize_rows=norm, precision="float32", fast_sqdists=True)
    try:
        algo.prepare_data(source_points=y, target_points=x)
        algo.fit()
        algo.prepare_query(source_signal=b)
        algo.query()
        got = algo.get_result()
        assert algo.device_kernel == "fastmm_kernel"
        algo.prepare_query(source_signal=2.0 * b)  # a new signal on the same points: only the signal operands are repacked
        algo.query()
        got2 = algo.get_result()
    finally:
        algo.done()
    want = kmvp_oracle.product(kernel="gaussian", source_points=y, target_points=x, source_signal=b, normalize_rows=norm)
    ref32 = kmvp_oracle.product(kernel="gaussian", source_points=y, target_points=x, source_signal=b, normalize_rows=norm,
                                precision=np.float32)
    assert got.shape == (n, E)
    col_err = np.abs(got - want).max(axis=0) / np.abs(want).max(axis=0)
    # per column, the rule of this file: the float32 tolerance or twice the error of the reference's own float32 arithmetic
    # (columns whose 2051 terms cancel to a hundredth of their mass show every rounding a hundred times larger)
    col_tol = np.maximum(TOL32, 2 * np.abs(ref32 - want).max(axis=0) / np.abs(want).max(axis=0))
    assert (col_err <= col_tol).all(), (col_err, col_tol)
    assert np.array_equal(got2, 2.0 * got)  # powers of two go through the column scales exactly


def test_cfastmm_inverse_distance_with_several_signal_columns(expected):
    """1/r with an (M, E) signal on the matrix cores (bruteforce.py:8-15 + :142-153): cfastmm_kernel with
    the per-target power-of-two scale folded into the target operand.  (1) the reference's golden same-points cases (E = 3,
    plain and normalised), every tile count, bitwise equal across tile counts; (2) 20000 points, D = 1 .. 4, E = 16 / 33,
    pairs 1e-6 apart and a DUPLICATED point -- its two rows must be non-finite as in the reference (coincident pair that is not
    the target's own index: 1/0), every other row finite and within the float32 rule; (3) what auto picks."""
    for name in ("inverse-distance-N193-M193-D3-E3-sp", "inverse-distance-N193-M193-D3-E3-nr-sp"):
        case = next(c for c in CASES if c["name"] == name)
        y, x, b = golden_cases.make_inputs(case)
        truth = expected[f"{name}/f64"]
        ref32 = expected[f"{name}/f32"].astype(np.float64)
        outs = []
        for tiles in (1, 2):
            got, extra = run_plugin(case, y, x, b, "float32", fast_sqdists="centred", fast_tiles=tiles)
            assert extra["device_kernel"] == "cfastmm_kernel" and "online shift" in extra["dispatch_note"], extra
            assert np.array_equal(row_finite(got), row_finite(truth))
            assert rel_err(got, truth) <= max(TOL32, 2 * rel_err(ref32, truth)), (name, tiles, rel_err(got, truth))
            outs.append(got)
        assert np.array_equal(outs[0], outs[1])
    rs = np.random.RandomState(31)
    for D, E, norm in ((3, 16, True), (3, 16, False), (1, 5, False), (2, 33, True), (4, 8, False)):
        n = 20000
        y = rs.rand(n, D)
        y[100] = y[7] + 1e-6          # a nearly coincident pair: 1/r = 1e6 beside sums of order 1e4
        y[4000] = y[12345]            # a duplicated point: rows 4000 and 12345 are 1/0 in the reference
        b = rs.randn(n, E)
        y32 = y.astype(np.float32).astype(np.float64)  # the arithmetic is checked on what float32 sees
        rows = np.concatenate([rs.choice(n, size=300, replace=False), [7, 100, 4000, 12345]])
        with np.errstate(all="ignore"):
            want = kmvp_oracle.product(kernel="inverse-distance", source_points=y32, source_signal=b, normalize_rows=norm, rows=rows)
            ref32 = kmvp_oracle.product(kernel="inverse-distance", source_points=y32, source_signal=b, normalize_rows=norm, rows=rows,
                                        precision=np.float32)
        got, extra = run_plugin(dict(kernel="inverse-distance", D=D, normalize_rows=norm), y32, None, b, "float32")
        assert extra["device_kernel"] == "cfastmm_kernel", extra
        fin = row_finite(want)
        assert not fin[-2:].any() and fin[:-2].sum() >= 300
        assert np.array_equal(row_finite(got[rows]), fin), (D, E, norm)
        ok = fin & row_finite(ref32)
        e, e32 = rel_err(got[rows][ok], want[ok]), rel_err(ref32[ok], want[ok])
        assert e <= max(TOL32, 2 * e32), (D, E, norm, e, e32)
    # targets != sources: the zero rule there is index-based without coincidence -- the centred forms do not apply
    x = rs.rand(500, 3)
    got, extra = run_plugin(dict(kernel="inverse-distance", D=3), y32[:3000, :3] if y32.shape[1] >= 3 else rs.rand(3000, 3), x,
                            rs.randn(3000, 16), "float32")
    assert extra["device_kernel"] == "lowd_kernel", extra


CFMM_MULTI = [c for c in CASES if c["D"] <= 4 and c["kernel"] in ("gaussian", "absolute-exponential") and c["E"] > 1
              and not c["density_estimation"]]


@pytest.mark.parametrize("case", CFMM_MULTI, ids=[c["name"] for c in CFMM_MULTI])
def test_cfastmm_kernel_with_several_signal_columns_matches_reference(case, expected):
    """E > 1, fast_sqdists="centred": cfast_kernel's distances (group centres, exact near pairs) feeding the second
    matrix product of fastmm_kernel (kmvp_cfastmm.hpp) -- exp(-r) and the Gaussian, D <= 4."""
    y, x, b = golden_cases.make_inputs(case)
    truth = expected[f"{case['name']}/f64"]
    ref32 = expected[f"{case['name']}/f32"].astype(np.float64)
    results = []
    for tiles in (1, 2):
        got, extra = run_plugin(case, y, x, b, "float32", fast_sqdists="centred", fast_tiles=tiles)
        assert extra["device_kernel"] == "cfastmm_kernel", extra
        tol = max(TOL32, 2 * rel_err(ref32, truth))
        assert got.shape == truth.shape and rel_err(got, truth) <= tol, (tiles, rel_err(got, truth), tol)
        results.append(got)
    assert np.array_equal(results[0], results[1])


@pytest.mark.parametrize("kernel,D,E,norm", [("absolute-exponential", 3, 40, True), ("absolute-exponential", 1, 5, False),
                                             ("absolute-exponential", 4, 33, False), ("gaussian", 2, 17, True)])
def test_cfastmm_kernel_near_pairs_column_blocks_and_wide_clouds(kernel, D, E, norm):
    """Clustered clouds far from the origin and wider than fast_kernel's radius rule, with exact and near duplicates
    between targets and sources (the exact branch of the group search must fire: exp(-r) has a kink at r = 0), ragged N
    and M, more than 32 columns; auto picks cfastmm_kernel here.  Against the float64 oracle on the float32-rounded
    inputs."""
    rng = np.random.RandomState(7 * D + E)
    n, m = 1531, 2777
    centres = rng.rand(12, D) * 9.0 + 100.0
    y = (centres[rng.randint(12, size=m)] + 0.05 * rng.randn(m, D)).astype(np.float32).astype(np.float64)
    x = (centres[rng.randint(12, size=n)] + 0.05 * rng.randn(n, D)).astype(np.float32).astype(np.float64)
    x[:200] = y[:200]                                                   # coincident pairs
    x[200:400] = (y[200:400] + 1e-5).astype(np.float32).astype(np.float64)  # pairs a few float32 ulps apart
    b = rng.randn(m, E).astype(np.float32).astype(np.float64)
    algo = MI355XProduct(kernel=kernel, dimension=D, normalize_rows=norm, precision="float32")
    try:
        algo.prepare_data(source_points=y, target_points=x)
        algo.fit()
        algo.prepare_query(source_signal=b)
        algo.query()
        got = algo.get_result()
        assert algo.device_kernel == "cfastmm_kernel", algo.device_kernel
    finally:
        algo.done()
    want = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b, normalize_rows=norm)
    ref32 = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b, normalize_rows=norm,
                                precision=np.float32)
    assert got.shape == (n, E)
    assert rel_err(got, want) <= max(TOL32, 2 * rel_err(ref32, want)), (rel_err(got, want), rel_err(ref32, want))


FMM_ABSEXP = [c for c in CASES if 5 <= c["D"] <= 64 and c["kernel"] == "absolute-exponential" and not c["density_estimation"]]


@pytest.mark.parametrize("case", FMM_ABSEXP, ids=[c["name"] for c in FMM_ABSEXP])
def test_fastmm_kernel_for_exp_of_minus_r_matches_reference(case, expected):
    """exp(-r) at 5 <= D <= 64 (one or many signal columns) on fastmm_kernel: the expansion around one centre with the
    closest pairs recomputed in the difference form (fast_sqdists=True forces it; auto takes it beyond D = 8 or four
    columns) -- held to the tolerance of the difference form."""
    y, x, b = golden_cases.make_inputs(case)
    truth = expected[f"{case['name']}/f64"]
    ref32 = expected[f"{case['name']}/f32"].astype(np.float64)
    results = []
    for tiles in (1, 2):
        got, extra = run_plugin(case, y, x, b, "float32", fast_sqdists=True, fast_tiles=tiles)
        if extra["device_kernel"] == "fast_kernel":
            return  # clouds outside the radius rule: the forced flag means the reference's plain expanded form there
        assert extra["device_kernel"] == "fastmm_kernel", extra
        tol = max(TOL32, 2 * rel_err(ref32, truth))
        assert got.shape == truth.shape and rel_err(got, truth) <= tol, (tiles, rel_err(got, truth), tol)
        results.append(got)
    assert np.array_equal(results[0], results[1])


@pytest.mark.parametrize("D,E,norm", [(8, 5, True), (16, 1, False), (16, 20, True), (33, 3, False), (64, 40, True)])
def test_fastmm_exp_of_minus_r_with_coincident_and_nearly_coincident_pairs(D, E, norm):
    """The exact branch: targets that coincide with sources, or sit a few float32 ulps from them, inside an otherwise
    ordinary cloud (exp(-r) has a kink at r = 0: the one-centre expansion alone would be off by ~5e-4 on such a pair),
    ragged sizes, auto dispatch.  Against the float64 oracle on the float32-rounded inputs."""
    rng = np.random.RandomState(3 * D + E)
    n, m = 1171, 1999
    y = (rng.rand(m, D) / np.sqrt(D / 3.0)).astype(np.float32).astype(np.float64)
    x = (rng.rand(n, D) / np.sqrt(D / 3.0)).astype(np.float32).astype(np.float64)
    x[:300] = y[:300]
    x[300:500] = (y[300:500] * (1.0 + 3e-7)).astype(np.float32).astype(np.float64)
    # few neighbours, large weights on the coincident sources: the near pairs dominate their rows
    b = rng.randn(m, E).astype(np.float32).astype(np.float64)
    b[:500] *= 50.0
    algo = MI355XProduct(kernel="absolute-exponential", dimension=D, normalize_rows=norm, precision="float32")
    try:
        algo.prepare_data(source_points=y, target_points=x)
        algo.fit()
        algo.prepare_query(source_signal=b)
        algo.query()
        got = algo.get_result()
        assert algo.device_kernel == "fastmm_kernel", algo.device_kernel
    finally:
        algo.done()
    want = kmvp_oracle.product(kernel="absolute-exponential", source_points=y, target_points=x, source_signal=b,
                               normalize_rows=norm)
    ref32 = kmvp_oracle.product(kernel="absolute-exponential", source_points=y, target_points=x, source_signal=b,
                                normalize_rows=norm, precision=np.float32)
    assert got.shape == (n, E)
    assert rel_err(got, want) <= max(TOL32, 2 * rel_err(ref32, want)), (rel_err(got, want), rel_err(ref32, want))
    # row by row on the rows of the near pairs (a relative error of 5e-4 on one dominant term would show here)
    rows = np.arange(500)
    row_err = np.abs(got[rows] - want[rows]).max(axis=1) / np.abs(want[rows]).max(axis=1)
    assert row_err.max() <= 2e-5, row_err.max()


def test_auto_choice_between_fastmm_and_cellmm():
    """auto takes the cheaper of the two matrix-core forms by the tile counts: at 1e5 uniform points two columns are
    cheaper as two cellmm_kernel launches, nine as one fastmm_kernel pass."""
    n = 100_000
    seen = {}
    for E in (2, 9):
        y, b = kmvp_oracle.uniform_cube(n, 3, E=E)
        algo = MI355XProduct(kernel="gaussian", dimension=3, precision="float32")
        try:
            algo.prepare_data(source_points=y, target_points=y, same_points=True)
            algo.fit()
            algo.prepare_query(source_signal=b)
            algo.query()
            got = algo.get_result()
            seen[E] = algo.device_kernel
        finally:
            algo.done()
        rows = np.random.RandomState(E).choice(n, size=64, replace=False)
        want = c_oracle.product(kernel="gaussian", source_points=y, source_signal=b, rows=rows)
        assert rel_err(got[rows], want) <= TOL32
    assert seen == {2: "cellmm_kernel", 9: "fastmm_kernel"}, seen


def test_low_d_attention_with_16_value_channels_at_1e5():
    """D = 3, E = 16, N = M = 1e5, row-normalised Gaussian attention under 5 ms -- picked up by
    fastmm_kernel by itself (17 columns in one pass, 2.4-2.6 ms on the device; 17 launches of cellmm_kernel: 12.0 ms,
    the column-blocked difference form: 12.9 ms), checked on 256 rows against the oracle."""
    n, E = 100_000, 16
    y, b = kmvp_oracle.uniform_cube(n, 3, E=E)
    algo = MI355XProduct(kernel="gaussian", dimension=3, normalize_rows=True, precision="float32")
    try:
        algo.prepare_data(source_points=y, target_points=y, same_points=True)
        algo.fit()
        algo.prepare_query(source_signal=b)
        algo.query()
        algo.query()
        got = algo.get_result()
        kname, ms = algo.device_kernel, algo.device_total_ms
    finally:
        algo.done()
    rows = np.random.RandomState(8).choice(n, size=256, replace=False)
    want = c_oracle.product(kernel="gaussian", source_points=y, source_signal=b, rows=rows, normalize_rows=True)
    assert kname == "fastmm_kernel", kname
    assert got.shape == (n, E) and rel_err(got[rows], want) <= TOL32, rel_err(got[rows], want)
    assert got.min() >= b.min() - 1e-4 and got.max() <= b.max() + 1e-4  # convex combinations of the signal rows
    print(f"D=3 E=16 N=M=1e5 normalised attention: {kname}, {E + 1} columns in one pass, {ms:.2f} ms on the device")
    assert ms < 5.0, ms


def test_golden_cases_reached_both_cell_kernels():
    assert {"cellmm_kernel", "cell_kernel", "E>1:cellmm_kernel"} <= CELL_KERNELS_SEEN, CELL_KERNELS_SEEN


def test_cell_kernel_shapes_offsets_and_auto_policy():
    """cell_kernel beyond the golden set: clouds with thousands of points per cell (several tiles per
    cell, several cells per wave, ragged last tiles), distinct target and source clouds of different
    extent, a cloud far from the origin, D = 1 and 2, normalised rows and density; the auto policy
    takes it when the clouds fill the cells (and not for a sparse cloud), with as many target tiles per
    wave as the padding allows."""
    rs = np.random.RandomState(2024)
    for case_no, (D, n, m, side, offset) in enumerate([(3, 70000, 70000, 0.25, 0.0), (3, 40000, 90000, 0.3, 1.0e3),
                                                        (2, 50000, 33000, 1.0, -5.0), (1, 33000, 40000, 3.0, 0.0),
                                                        (3, 40000, 40000, 1.0, 0.0)]):
        same = case_no in (0, 4)
        y = (rs.rand(m, D) * side + offset).astype(np.float32)
        x = None if same else (rs.rand(n, D) * side * 0.8 + offset).astype(np.float32)
        b = rs.randn(m, 1).astype(np.float32)
        norm = case_no == 1
        dens = case_no == 2
        rows = rs.choice(m if same else n, size=300, replace=False)
        tx = (y if same else x)[rows].astype(np.float64)
        want = kmvp_oracle.product(kernel="gaussian", source_points=y.astype(np.float64), target_points=tx,
                                   source_signal=None if dens else b.astype(np.float64), normalize_rows=norm,
                                   density_estimation=dens)
        case = dict(kernel="gaussian", D=D, normalize_rows=norm)
        cell_name = "cellmm_kernel"  # also for normalised rows (a second launch with b = 1 for the denominator)
        got, extra = run_plugin(case, y, x, None if dens else b, "float32", fast_sqdists="cells")
        assert extra["device_kernel"] == cell_name, extra
        assert rel_err(got[rows], want) <= TOL32, (case_no, rel_err(got[rows], want))
        valu, extra = run_plugin(case, y, x, None if dens else b, "float32", fast_sqdists="cells-valu")
        assert extra["device_kernel"] == "cell_kernel", extra
        assert rel_err(valu[rows], want) <= TOL32, (case_no, rel_err(valu[rows], want))
        auto, extra = run_plugin(case, y, x, None if dens else b, "float32")
        # clouds that fill the cells go to the cell kernels by themselves; 4e4 points in the unit cube
        # (40 per cell of side 0.103: more than 30 % of the tile slots would be padding) stay with fast_kernel
        assert extra["device_kernel"] == ("fast_kernel" if case_no == 4 else cell_name), (case_no, extra)
        # the caller is told WHY the fastest form was not taken (kmvp_last_dispatch_note), and nothing when it was
        assert ("too few points per grid cell" in extra["dispatch_note"]) == (case_no == 4), extra
        assert rel_err(auto[rows], want) <= TOL32, (case_no, rel_err(auto[rows], want))


CELLS64 = CELLS


@pytest.mark.parametrize("case", CELLS64, ids=[c["name"] for c in CELLS64])
def test_cell64_kernel_matches_reference(case, expected):
    """float64 cell form (kmvp_cell64.hpp: exp() range-reduced by grid cells, degree-8 remainder): held to the
    float64 tolerance on the golden Gaussian cases."""
    y, x, b = golden_cases.make_inputs(case)
    want = expected[f"{case['name']}/f64"]
    got, extra = run_plugin(case, y, x, b, np.float64, fast_sqdists="cells")
    if not (case["normalize_rows"] and case["density_estimation"]):
        assert extra["device_kernel"] == "cell64_kernel"
    assert rel_err(got, want) <= TOL64, rel_err(got, want)


def test_cell64_kernel_at_scale_and_in_the_solver():
    """1e5 points (BASELINE config 5's shape): the auto rule takes the float64 cell form, the product matches
    the oracle on a row subset for x == y and x != y, clustered clouds included, and conjugate gradients on
    that operator reach the residual."""
    rs = np.random.RandomState(64)
    n = 100_000
    y, b = kmvp_oracle.uniform_cube(n, 3)
    rows = rs.choice(n, size=200, replace=False)
    want = kmvp_oracle.product(kernel="gaussian", source_points=y, target_points=y[rows], source_signal=b)
    got, extra = run_plugin(dict(kernel="gaussian", D=3), y, None, b, np.float64)
    assert extra["device_kernel"] == "cell64_kernel"
    assert rel_err(got[rows], want) <= TOL64, rel_err(got[rows], want)
    # distinct clouds, clusters and duplicates, D = 2
    y2 = np.concatenate([rs.randn(30000, 2) * 0.01 + 0.3, rs.rand(20000, 2), np.tile(rs.rand(1, 2), (500, 1))])
    x2 = np.concatenate([rs.rand(33000, 2) * 1.3 - 0.1, y2[:777]])
    b2 = rs.randn(len(y2), 1)
    rows = rs.choice(len(x2), size=200, replace=False)
    want = kmvp_oracle.product(kernel="gaussian", source_points=y2, target_points=x2[rows], source_signal=b2)
    got, extra = run_plugin(dict(kernel="gaussian", D=2), y2, x2, b2, np.float64, fast_sqdists="cells")
    assert extra["device_kernel"] == "cell64_kernel"
    assert rel_err(got[rows], want) <= TOL64, rel_err(got[rows], want)
    want = kmvp_oracle.product(kernel="gaussian", source_points=y2, target_points=x2[rows], source_signal=b2, normalize_rows=True)
    got, extra = run_plugin(dict(kernel="gaussian", D=2, normalize_rows=True), y2, x2, b2, np.float64, fast_sqdists="cells")
    assert extra["device_kernel"] == "cell64_kernel"
    assert rel_err(got[rows], want) <= TOL64, rel_err(got[rows], want)
    # the solver on that operator (x := K b0, then solve K b = x); 80000 points fill the 128-target tiles of the
    # 216 cells (370 per cell), 40000 would leave 38 % of the slots empty and stay with the difference form
    m = 80000
    ys = rs.rand(m, 3)
    b0 = rs.randn(m, 1)
    algo = MI355XProduct(kernel="gaussian", dimension=3, precision=np.float64)
    try:
        algo.prepare_data(source_points=ys, target_points=ys, same_points=True)
        algo.prepare_query(source_signal=b0)
        algo.query()
        a = algo.get_result()
        assert algo.get_additional()["device_kernel"] == "cell64_kernel"
    finally:
        algo.done()
    solver = MI355XSolver(kernel="gaussian", dimension=3, precision=np.float64, rtol=1e-6, maxit=3000)
    try:
        solver.prepare_data(source_points=ys)
        solver.fit()
        solver.prepare_query(target_signal=a)
        solver.query()
        extra = solver.get_additional()
        assert extra["cg_converged"] and extra["cg_relative_residual"] <= 1e-6, extra
        assert solver._ctx.last_kernel_name == "cell64_kernel"
        true = np.linalg.norm(c_oracle.product(kernel="gaussian", source_points=ys, source_signal=solver.get_result(),
                                               rows=np.arange(256)) - a[:256]) / np.linalg.norm(a[:256])
        assert true <= 1e-5, true  # the residual holds against the oracle's operator too (256 rows)
    finally:
        solver.done()


def test_cell_kernel_on_clustered_clouds():
    """Cell occupancies from one point to tens of thousands (tight clusters on a thin uniform
    background, duplicated points, a cluster exactly on a cell boundary): the tile lists, the padding to
    TT tiles and the skipping of empty tiles must not depend on how full a cell is."""
    rs = np.random.RandomState(77)
    h = np.sqrt(2 * 0.016 / 3)
    blobs = [rs.randn(30000, 3) * 0.004 + 0.31, rs.randn(5000, 3) * 0.02 + np.array([0.7, 0.2, 0.55]),
             np.tile(rs.rand(1, 3), (700, 1)),                       # 700 copies of one point
             np.floor(rs.rand(1, 3) * 5) * h + rs.randn(3000, 3) * 1e-4,   # straddles cell faces of the grid
             rs.rand(1500, 3)]
    y = np.concatenate(blobs).astype(np.float32)
    rs.shuffle(y)
    b = rs.randn(len(y), 1).astype(np.float32)
    x = np.concatenate([y[:20000] + np.float32(1e-3), rs.rand(777, 3).astype(np.float32)])
    rows = rs.choice(len(x), size=400, replace=False)
    want = kmvp_oracle.product(kernel="gaussian", source_points=y.astype(np.float64), target_points=x[rows].astype(np.float64),
                               source_signal=b.astype(np.float64))
    for form, kname in (("cells", "cellmm_kernel"), ("cells-valu", "cell_kernel")):
        for tiles in (1, 4, 8):
            got, extra = run_plugin(dict(kernel="gaussian", D=3), y, x, b, "float32", fast_sqdists=form, fast_tiles=tiles)
            assert extra["device_kernel"] == kname
            assert rel_err(got[rows], want) <= TOL32, (form, tiles, rel_err(got[rows], want))
    same, extra = run_plugin(dict(kernel="gaussian", D=3, normalize_rows=True), y, None, b, "float32", fast_sqdists="cells")
    rows = rs.choice(len(y), size=400, replace=False)
    want = kmvp_oracle.product(kernel="gaussian", source_points=y.astype(np.float64), target_points=y[rows].astype(np.float64),
                               source_signal=b.astype(np.float64), normalize_rows=True)
    assert extra["device_kernel"] == "cellmm_kernel" and rel_err(same[rows], want) <= TOL32, rel_err(same[rows], want)
    same, extra = run_plugin(dict(kernel="gaussian", D=3, normalize_rows=True), y, None, b, "float32", fast_sqdists="cells-valu")
    assert extra["device_kernel"] == "cell_kernel" and rel_err(same[rows], want) <= TOL32, rel_err(same[rows], want)


def test_cell_paths_follow_new_points_and_signals_in_one_context():
    """One context, points and signals replaced in turn (float32 and float64 cell paths): the cell order belongs
    to a points version, the source image to a signal version -- neither may be reused across an upload."""
    rs = np.random.RandomState(31)
    for dtype, npdt, tol, kname in ((_lib.KMVP_F32, np.float32, TOL32, "cellmm_kernel"), (_lib.KMVP_F32, np.float32, TOL32, "cell_kernel"),
                                    (_lib.KMVP_F64, np.float64, TOL64, "cell64_kernel")):
        ctx = _lib.Context(0)
        try:
            ctx.set_option("fast_sqdists", 4 if kname == "cell_kernel" else 3)
            for n, side in ((40000, 0.4), (36000, 0.7), (40000, 0.4)):
                y = (rs.rand(n, 3) * side).astype(npdt)
                ctx.set_points(y, None, dtype)
                rows = rs.choice(n, size=100, replace=False)
                for _ in range(2):
                    b = rs.randn(n, 1).astype(npdt)
                    ctx.set_signal(b)
                    ctx.run("gaussian", False)
                    got = ctx.get_result(n, 1)
                    assert ctx.last_kernel_name == kname
                    want = kmvp_oracle.product(kernel="gaussian", source_points=y.astype(np.float64),
                                               target_points=y[rows].astype(np.float64), source_signal=b.astype(np.float64))
                    assert rel_err(got[rows], want) <= tol, (kname, n, rel_err(got[rows], want))
        finally:
            ctx.close()


def test_fast_sqdists_auto_policy():
    """auto: unit-cube gaussian -> matrix cores; same cloud blown up 100x, or 1/r -> difference form."""
    y, b = kmvp_oracle.uniform_cube(2000, 3)
    _, extra = run_plugin(dict(kernel="gaussian", D=3), y, None, b, "float32")
    assert extra["device_kernel"] == "fast_kernel"
    _, extra = run_plugin(dict(kernel="gaussian", D=3), 100 * y + 1e4, None, b, "float32")
    assert extra["device_kernel"] == "cfast_kernel"  # large scaled radius: per-group centres
    _, extra = run_plugin(dict(kernel="inverse-distance", D=3), y, None, b, "float32")
    assert extra["device_kernel"] == "cfast_kernel"  # same points: zero rule = coincident pairs
    _, extra = run_plugin(dict(kernel="inverse-distance", D=3), y, y[:1500] + 0.5, b, "float32")
    assert extra["device_kernel"] == "lowd_kernel"   # index-based zero rule on distinct points
    _, extra = run_plugin(dict(kernel="absolute-exponential", D=3), y, None, b, "float32")
    assert extra["device_kernel"] == "cfast_kernel"
    y5 = np.random.RandomState(0).rand(500, 5)
    _, extra = run_plugin(dict(kernel="absolute-exponential", D=5), y5, None, b[:500], "float32")
    assert extra["device_kernel"] == "lowd_kernel"   # D > 4
    _, extra = run_plugin(dict(kernel="gaussian", D=3), y, None, b, "float32", fast_sqdists=False)
    assert extra["device_kernel"] == "lowd_kernel"
    got, extra = run_plugin(dict(kernel="gaussian", D=3), y + 1e3, None, b, "float32")  # far from the origin
    assert extra["device_kernel"] == "fast_kernel"  # centring makes the offset harmless
    want = kmvp_oracle.product(kernel="gaussian", source_points=(y + 1e3).astype(np.float32).astype(np.float64),
                               source_signal=b)
    assert rel_err(got, want) <= TOL32


def test_clusters_far_apart_and_non_finite_points():
    """Coordinates are never centred or scaled before a difference is formed: two unit clusters
    1e4 apart (bounding-box midpoint far from both) keep the accuracy of the difference form of
    the reference on the same float32 inputs; non-finite points behave as in the reference
    (a source at infinity contributes exp(-inf) = 1/inf = 0, a NaN target gives a NaN row)."""
    rs = np.random.RandomState(11)
    n = 4000
    y = rs.rand(n, 3)
    y[n // 2:] += 1e4
    b = rs.randn(n, 1)
    y32 = y.astype(np.float32).astype(np.float64)  # what every float32 backend is handed
    for kernel in golden_cases.KERNELS:
        want = kmvp_oracle.product(kernel=kernel, source_points=y32, source_signal=b)
        for opts in (dict(), dict(fast_sqdists=False)):
            got, extra = run_plugin(dict(kernel=kernel, D=3), y, None, b, "float32", **opts)
            assert extra["device_kernel"] == ("lowd_kernel" if opts else "cfast_kernel")
            assert rel_err(got, want) <= TOL32, (kernel, extra["device_kernel"], rel_err(got, want))

    # mid-dimensional cloud far from the origin: every one of the D bounding-box centres has to be right
    # for the global-centre expansion (D = 20 > 8 once overran the layout of the centre buffer)
    yd = rs.rand(2000, 20) / np.sqrt(20.0) + 1.0e3
    bd = rs.randn(2000, 1)
    yd32 = yd.astype(np.float32).astype(np.float64)
    want = kmvp_oracle.product(kernel="gaussian", source_points=yd32, source_signal=bd)
    got, extra = run_plugin(dict(kernel="gaussian", D=20), yd, None, bd, "float32")
    assert extra["device_kernel"] == "fast_kernel"
    assert rel_err(got, want) <= TOL32, rel_err(got, want)

    y = rs.rand(1500, 3)
    y[7] = np.inf
    x = rs.rand(700, 3)
    x[5, 1] = np.nan
    b = rs.randn(1500, 1)
    for kernel in golden_cases.KERNELS:
        with np.errstate(all="ignore"):
            want = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b)
        assert np.isnan(want[5]).all() and np.isfinite(np.delete(want, 5, axis=0)).all()
        for opts in (dict(), dict(fast_sqdists=False), dict(fast_sqdists="centred"), dict(fast_sqdists="cells")):
            if kernel == "inverse-distance" and opts.get("fast_sqdists") == "centred":
                continue  # index-based zero rule on distinct points: difference form only
            got, extra = run_plugin(dict(kernel=kernel, D=3), y, x, b, "float32", **opts)
            assert np.isnan(got[5]).all(), (kernel, extra["device_kernel"])
            assert rel_err(np.delete(got, 5, axis=0), np.delete(want, 5, axis=0)) <= TOL32, (kernel, extra["device_kernel"])


def test_non_finite_points_with_several_signal_columns():
    """The same non-finite inputs through the multi-column matrix-core kernels (a source at infinity: zero contribution to
    every column; a NaN target coordinate: a NaN row), plain and normalised, against the oracle."""
    rs = np.random.RandomState(12)
    y = rs.rand(1500, 3)
    y[7] = np.inf
    x = rs.rand(700, 3)
    x[5, 1] = np.nan
    b = rs.randn(1500, 6)
    for kernel in ("gaussian", "absolute-exponential"):
        for norm in (False, True):
            with np.errstate(all="ignore"):
                want = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b,
                                           normalize_rows=norm)
            assert np.isnan(want[5]).all() and np.isfinite(np.delete(want, 5, axis=0)).all()
            # (not fast_sqdists=True: the one-centre expansion turns an infinite point into inf - inf for every pair, as
            # the reference's own fast form does, bruteforce.py:36-49)
            for opts in (dict(), dict(fast_sqdists="centred")):
                got, extra = run_plugin(dict(kernel=kernel, D=3, normalize_rows=norm), y, x, b, "float32", **opts)
                assert extra["device_kernel"] in ("fastmm_kernel", "cfastmm_kernel"), extra
                assert np.isnan(got[5]).all(), (kernel, norm, extra["device_kernel"])
                assert rel_err(np.delete(got, 5, axis=0), np.delete(want, 5, axis=0)) <= TOL32, (kernel, norm, extra["device_kernel"])


def test_random_shapes_flags_and_precisions():
    """Seeded sweep over shapes the golden set does not hold: every dispatch decision (difference /
    global-centre / per-group-centre / generic kernels, small-problem launch shapes, ragged tiles,
    x == y and x != y, all query() branches) against the numpy oracle."""
    rs = np.random.RandomState(20240607)
    for case_no in range(72):
        kernel = golden_cases.KERNELS[case_no % 3]
        D = int(rs.choice([1, 2, 3, 3, 3, 4, 5, 8, 9, 12, 17, 23, 24, 39, 40]))
        E = int(rs.choice([1, 1, 1, 2, 4, 5]))
        M = int(rs.choice([1, 31, 33, 127, 129, 257, 700, 1500]))
        same = bool(rs.rand() < 0.5)
        N = M if same else int(rs.choice([1, 32, 65, 255, 513, 900]))
        norm = bool(rs.rand() < 0.4)
        dens = bool(rs.rand() < 0.2)
        prec = "float64" if rs.rand() < 0.35 else "float32"
        spread = float(rs.choice([1.0, 1.0, 6.0]))  # 6: scaled radius beyond the global-centre rule
        scale = spread / np.sqrt(max(D, 3) / 3.0)  # typical squared distances stay O(spread^2): no float32 underflow
        y = rs.rand(M, D) * scale
        x = None if same else rs.rand(N, D) * scale
        b = None if dens else rs.randn(M, E)
        if prec == "float32":  # the backend is handed float32 inputs: the oracle gets the same numbers
            y = y.astype(np.float32).astype(np.float64)
            x = None if x is None else x.astype(np.float32).astype(np.float64)
            b = None if b is None else b.astype(np.float32).astype(np.float64)
        want = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b,
                                   normalize_rows=norm, density_estimation=dens)
        case = dict(kernel=kernel, D=D, normalize_rows=norm)
        got, extra = run_plugin(case, y, x, b, prec)
        assert got.shape == want.shape, (case_no, got.shape, want.shape)
        fin = np.isfinite(want).all(axis=-1)
        assert np.array_equal(np.isfinite(got).all(axis=-1), fin), (case_no, kernel, extra)
        tol = TOL64 if prec == "float64" else 2e-5
        assert rel_err(got, want) <= tol, (case_no, kernel, D, E, N, M, same, norm, dens, prec, extra["device_kernel"],
                                          rel_err(got, want))


def test_dimensions_beyond_128():
    """D > 128 (lowd_big_kernel: rows padded to multiples of 32 coordinates, chunked walk): ragged D and
    E, every query() branch, both precisions, batches of sources that end inside a segment."""
    rs = np.random.RandomState(129)
    shapes = [(129, 1), (160, 3), (200, 9), (257, 8), (300, 17), (1000, 1)]
    for case_no, (D, E) in enumerate(shapes * 2):
        kernel = golden_cases.KERNELS[case_no % 3]
        M = int(rs.choice([1, 7, 9, 333, 1030]))
        same = bool(case_no % 2)
        N = M if same else int(rs.choice([1, 65, 300]))
        norm = bool(case_no % 3 == 1)
        dens = bool(case_no % 5 == 4)
        prec = "float64" if case_no >= len(shapes) else "float32"
        scale = 1.0 / np.sqrt(D / 3.0)
        y = rs.rand(M, D) * scale
        x = None if same else rs.rand(N, D) * scale
        b = None if dens else rs.randn(M, E)
        if prec == "float32":
            y = y.astype(np.float32).astype(np.float64)
            x = None if x is None else x.astype(np.float32).astype(np.float64)
            b = None if b is None else b.astype(np.float32).astype(np.float64)
        want = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b,
                                   normalize_rows=norm, density_estimation=dens)
        got, extra = run_plugin(dict(kernel=kernel, D=D, normalize_rows=norm), y, x, b, prec)
        if not (norm and dens):  # normalised density is all ones without a launch
            assert extra["device_kernel"] == "lowd_big_kernel", extra
        assert got.shape == want.shape
        tol = TOL64 if prec == "float64" else 2e-5
        assert rel_err(got, want) <= tol, (case_no, kernel, D, E, N, M, same, norm, dens, prec, rel_err(got, want))


def test_random_options_never_break_a_product():
    """Seeded sweep over the tuning options (segments, tiles per wave, forced squared-distance forms,
    feeds): options that do not apply fall back silently, every combination returns the right sums
    (the forced global-centre form is only held to its own accuracy class for the Gaussian)."""
    rs = np.random.RandomState(777)
    for case_no in range(60):
        kernel = golden_cases.KERNELS[case_no % 3]
        D = int(rs.choice([1, 3, 3, 4, 6, 8, 12]))
        M = int(rs.choice([40, 257, 1000, 3000]))
        same = bool(rs.rand() < 0.6)
        N = M if same else int(rs.choice([33, 300, 2000]))
        norm = bool(rs.rand() < 0.3)
        y = rs.rand(M, D)
        x = None if same else rs.rand(N, D)
        b = rs.randn(M, 1)
        opts = {}
        if rs.rand() < 0.7:
            opts["fast_sqdists"] = [None, False, True, "centred"][int(rs.randint(4))]
        if rs.rand() < 0.5:
            opts["segments"] = int(rs.choice([1, 2, 3, 8, 17]))
        if rs.rand() < 0.5:
            opts["fast_tiles"] = int(rs.choice([1, 2, 4]))
        if rs.rand() < 0.3:
            opts["targets_per_lane"] = int(rs.choice([1, 2]))
        if rs.rand() < 0.3:
            opts["feed"] = int(rs.choice([0, 1]))
        # (T, feed) pairs that are only instantiated for the headline shape fall back inside the library
        if kernel == "gaussian" and D <= 3 and case_no % 2:
            opts["fast_sqdists"] = "cells"  # (drawn outside the seeded stream above, which predates this form)
        y32 = y.astype(np.float32).astype(np.float64)
        x32 = None if x is None else x.astype(np.float32).astype(np.float64)
        b32 = b.astype(np.float32).astype(np.float64)
        want = kmvp_oracle.product(kernel=kernel, source_points=y32, target_points=x32, source_signal=b32,
                                   normalize_rows=norm)
        got, extra = run_plugin(dict(kernel=kernel, D=D, normalize_rows=norm), y, x, b, "float32", **opts)
        forced_global = opts.get("fast_sqdists") is True and extra["device_kernel"] == "fast_kernel"
        tol = 2e-5 if (not forced_global or kernel == "gaussian") else 5e-2
        assert rel_err(got, want) <= tol, (case_no, kernel, D, N, M, same, norm, opts, extra["device_kernel"], rel_err(got, want))


def test_matrix_core_kernels_reproducible_and_tile_count_independent():
    """LDS-DMA staged kernels must not depend on timing: bitwise identical results run to run at a
    size where every CU is busy, and the same sums (to float32 rounding) whatever the number of
    target tiles per wave."""
    n = 300_000
    y, b = kmvp_oracle.uniform_cube(n, 3)
    y32, b32 = y.astype(np.float32), b.astype(np.float32)

    def product(kernel, fast, tiles=0, norm=False):
        ctx = _lib.Context(0)
        try:
            ctx.set_option("fast_sqdists", fast)
            if tiles:
                ctx.set_option("fast_tiles", tiles)
            ctx.set_points(y32, None, _lib.KMVP_F32)
            ctx.set_signal(b32)
            outs = []
            for _ in range(3):
                ctx.run(kernel, norm)
                outs.append(ctx.get_result(n, 1))
            name = ctx.last_kernel_name
        finally:
            ctx.close()
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), (kernel, name, tiles)
        return outs[0], name

    for kernel, fast, kname in (("gaussian", 1, "fast_kernel"), ("inverse-distance", 2, "cfast_kernel"),
                                ("absolute-exponential", 2, "cfast_kernel"), ("gaussian", 2, "cfast_kernel"),
                                ("gaussian", 3, "cellmm_kernel"), ("gaussian", 4, "cell_kernel")):
        base, name = product(kernel, fast, 1)
        assert name == kname
        scale = np.max(np.abs(base))
        for tiles in (2, 4, 8):  # 8: the cell kernels only, the others clamp to 4
            other, _ = product(kernel, fast, tiles)
            assert np.max(np.abs(other - base)) <= 5e-6 * scale, (kernel, kname, tiles, np.max(np.abs(other - base)) / scale)

    m, D, E = 16384, 64, 64
    rs = np.random.RandomState(m + D)
    yd = (rs.rand(m, D) / np.sqrt(D)).astype(np.float32)
    bd = rs.randn(m, E).astype(np.float32)
    for T in (0, 1, 2):  # pipelined / one / two target tiles per wave
        ctx = _lib.Context(0)
        try:
            if T:
                ctx.set_option("targets_per_lane", T)
            ctx.set_points(yd, None, _lib.KMVP_BF16)
            ctx.set_signal(bd)
            outs = []
            for _ in range(3):
                ctx.run("absolute-exponential", True)
                outs.append(ctx.get_result(m, E))
        finally:
            ctx.close()
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), T


TOL_BF16 = 1e-2  # bf16 inputs (8-bit mantissa) with fp32 accumulation; measured 2.4e-3 .. 4e-3
HIGH_D = [c for c in CASES if c["D"] >= 16]


@pytest.mark.parametrize("case", HIGH_D, ids=[c["name"] for c in HIGH_D])
def test_bfloat16_mfma_matches_reference(case, expected):
    y, x, b = golden_cases.make_inputs(case)
    truth = expected[f"{case['name']}/f64"]
    got, extra = run_plugin(case, y, x, b, "bfloat16")
    assert got.shape == truth.shape and extra["device_kernel"] in ("mfma_kernel", "mfma_pipe_kernel")
    assert rel_err(got, truth) <= TOL_BF16, rel_err(got, truth)
    # the non-pipelined instantiations (one / two target tiles per wave) compute the same sums
    for tiles in (1, 2):
        other, extra = run_plugin(case, y, x, b, "bfloat16", targets_per_lane=tiles)
        assert extra["device_kernel"] == "mfma_kernel"
        assert rel_err(other, truth) <= TOL_BF16
        assert rel_err(other, got) <= 1e-5, (tiles, rel_err(other, got))


def test_config3_attention_65536_row_subset():
    """BASELINE config 3: exponential-kernel attention, D = 64, E = 64, N = M = 65536, bf16 MFMA."""
    n, D, E = 65536, 64, 64
    rs = np.random.RandomState(n + D)
    y = rs.rand(n, D) / np.sqrt(D)  # points scaled by 1/sqrt(D)
    b = rs.randn(n, E)
    rows = np.random.RandomState(0).choice(n, size=128, replace=False)
    for kernel in ("absolute-exponential", "gaussian"):
        algo = MI355XProduct(kernel=kernel, dimension=D, normalize_rows=True, precision="bfloat16")
        try:
            algo.prepare_data(source_points=y, target_points=y, same_points=True)
            algo.prepare_query(source_signal=b)
            algo.query()
            a = algo.get_result()
        finally:
            algo.done()
        want = c_oracle.product(kernel=kernel, source_points=y, source_signal=b, normalize_rows=True, rows=rows)
        assert rel_err(a[rows], want) <= TOL_BF16
        assert a.min() >= b.min() - 0.05 and a.max() <= b.max() + 0.05  # convex combinations


def test_config3_attention_65536_with_contrast():
    """The C3 shape (N = M = 65536, D = 64, E = 64, exp(-r), row-normalised, bf16) on a cloud that DISCRIMINATES: in the
    bench's uniform cloud / sqrt(D) all distances are 0.41 +- 0.03, the weights differ by +-3 % and every row is close to
    the column mean of b -- a kernel whose weights were 10 % off would pass at 1e-2.  Here 64
    clusters of 1024 points (cluster radius ~0.25, centres up to ~4 apart) give distances from 0 to ~4: within a row the
    weights span e^0 ... e^-4 and the answer is dominated by the target's own cluster, so the value rows of that cluster
    -- not the global mean -- must come out.  128 rows against the float64 C oracle at the bf16 tolerance; every variant
    of the pipelined kernel and the plain kernel."""
    n, D, E = 65536, 64, 64
    rs = np.random.RandomState(n + D + 1)
    centres = rs.randn(64, D) * (2.0 / np.sqrt(D))          # |c - c'| ~ 2.8
    y = np.repeat(centres, n // 64, axis=0) + rs.randn(n, D) * (0.25 / np.sqrt(D))
    b = rs.randn(n, E) + np.repeat(rs.randn(64, E) * 3.0, n // 64, axis=0)  # value rows differ by cluster
    perm = rs.permutation(n)
    y, b = y[perm], b[perm]
    rows = np.random.RandomState(1).choice(n, size=128, replace=False)
    want = c_oracle.product(kernel="absolute-exponential", source_points=y, source_signal=b, normalize_rows=True, rows=rows)
    r = np.sqrt(((y[rows[:8], None, :] - y[None, ::64, :]) ** 2).sum(-1))
    assert r.min() < 0.6 and r.max() > 3.0, (r.min(), r.max())  # the weights of a row span more than e^-2.4
    col_mean = b.mean(axis=0)
    assert rel_err(np.tile(col_mean, (128, 1)), want) > 0.2  # "every row = the mean" is far outside the tolerance here
    seen = []
    for opts in (dict(), dict(mfma_variant=0), dict(mfma_variant=1), dict(mfma_variant=5), dict(targets_per_lane=2)):
        algo = MI355XProduct(kernel="absolute-exponential", dimension=D, normalize_rows=True, precision="bfloat16")
        try:
            algo.prepare_data(source_points=y, target_points=y, same_points=True)
            algo.set_query_arguments(**opts)
            algo.prepare_query(source_signal=b)
            algo.query()
            a = algo.get_result()
            seen.append(algo.device_kernel)
        finally:
            algo.done()
        assert rel_err(a[rows], want) <= TOL_BF16, (opts, rel_err(a[rows], want))
    assert seen[0] == "mfma_pipe_kernel" and seen[-1] == "mfma_kernel", seen


def test_every_tuning_variant_gives_the_same_answer(expected):
    case = next(c for c in CASES if c["name"] == "gaussian-N257-M193-D3-E1")
    y, x, b = golden_cases.make_inputs(case)
    want = expected[f"{case['name']}/f64"]
    for kernel in golden_cases.KERNELS:
        c = dict(case, kernel=kernel)
        want = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b)
        for feed in (0, 1):
            for T in (1, 2, 4, 8):
                for seg in (0, 1, 3, 8):
                    got, _ = run_plugin(c, y, x, b, np.float32, feed=feed if feed else None,
                                        targets_per_lane=T, segments=seg, fast_sqdists=False)
                    assert rel_err(got, want) <= TOL32, (kernel, feed, T, seg, rel_err(got, want))
        got, _ = run_plugin(c, y, x, b, np.float32, chunk=8, fast_sqdists=False)
        assert rel_err(got, want) <= TOL32
        if kernel == "gaussian":
            for tiles in (1, 2, 4):
                got, _ = run_plugin(c, y, x, b, np.float32, fast_sqdists=True, fast_tiles=tiles, segments=3)
                assert rel_err(got, want) <= TOL32


def test_results_are_bitwise_reproducible():
    y, b = kmvp_oracle.uniform_cube(5000, 3)
    case = dict(kernel="gaussian", D=3)
    a1, _ = run_plugin(case, y, None, b, np.float32)
    a2, _ = run_plugin(case, y, None, b, np.float32)
    assert np.array_equal(a1, a2)  # no atomics: fixed summation order


def test_source_shard_with_global_offset_matches_oracle():
    """What one rank computes when the sources are sharded (kmvp_set_points j_offset / M_total)."""
    case = dict(N=300, M=257, D=3, E=2, seed=11, same_points=False, density_estimation=False)
    y, x, b = golden_cases.make_inputs(case)
    for kernel in golden_cases.KERNELS:
        total = np.zeros((300, 2))
        for lo, hi in ((0, 130), (130, 257)):
            ctx = _lib.Context(0)
            try:
                ctx.set_option("partial_shard", 1)  # this shard's partial sums, on purpose: added up below
                ctx.set_points(np.ascontiguousarray(y[lo:hi]), x, _lib.KMVP_F64, j_offset=lo, M_total=257)
                ctx.set_signal(np.ascontiguousarray(b[lo:hi]))
                ctx.run(kernel, False)
                part = ctx.get_result(300, 2)
            finally:
                ctx.close()
            want, _ = kmvp_oracle.product(kernel=kernel, source_points=y[lo:hi], target_points=x,
                                          source_signal=b[lo:hi], j_offset=lo, M_total=257, raw_sums=True)
            assert rel_err(part, want) <= TOL64, (kernel, lo)
            total += part
        full = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b)
        assert rel_err(total, full) <= TOL64


def test_source_shards_of_the_multi_column_kernels_add_up():
    """The same for float32 products with several signal columns (fastmm_kernel / cfastmm_kernel; every rank scales the
    columns of ITS shard of the signal by its own powers of two): three uneven source shards against the whole product."""
    rs = np.random.RandomState(21)
    n, m, E = 700, 1300, 6
    y = rs.rand(m, 3).astype(np.float32)
    x = rs.rand(n, 3).astype(np.float32)
    b = (rs.randn(m, E) * 10.0 ** rs.randint(-3, 4, size=E)).astype(np.float32)
    for kernel, fast, kname in (("gaussian", 1, "fastmm_kernel"), ("absolute-exponential", 2, "cfastmm_kernel")):
        total = np.zeros((n, E))
        for lo, hi in ((0, 33), (33, 700), (700, m)):
            ctx = _lib.Context(0)
            try:
                ctx.set_option("partial_shard", 1)
                ctx.set_option("fast_sqdists", fast)
                ctx.set_points(np.ascontiguousarray(y[lo:hi]), x, _lib.KMVP_F32, j_offset=lo, M_total=m)
                ctx.set_signal(np.ascontiguousarray(b[lo:hi]))
                ctx.run(kernel, False)
                total += ctx.get_result(n, E)
                assert ctx.last_kernel_name == kname
            finally:
                ctx.close()
        full = kmvp_oracle.product(kernel=kernel, source_points=y.astype(np.float64), target_points=x.astype(np.float64),
                                   source_signal=b.astype(np.float64))
        col_err = np.abs(total - full).max(axis=0) / np.abs(full).max(axis=0)
        assert col_err.max() <= TOL32, (kernel, col_err)


def test_single_rank_rccl_communicator():
    """kmvp_comm_get_unique_id / kmvp_comm_init / ncclAllReduce with world == 1 on the one GPU.
    With a communicator attached every path goes through the exchange in the canonical unpadded
    layout [column][N] (ranks may pick kernels with different tile padding): lowd / fast / cfast /
    cell / mfma kernels, N not a multiple of any tile size."""
    n = 1000 + 37
    y, b = kmvp_oracle.uniform_cube(n, 3)
    cases = [("gaussian", True, _lib.KMVP_F32, 0, "lowd_kernel"), ("gaussian", True, _lib.KMVP_F32, 1, "fast_kernel"),
             ("inverse-distance", False, _lib.KMVP_F32, 2, "cfast_kernel"),
             ("gaussian", False, _lib.KMVP_F32, 3, "cellmm_kernel"), ("gaussian", False, _lib.KMVP_F32, 4, "cell_kernel"),
             ("gaussian", True, _lib.KMVP_F32, 3, "cellmm_kernel"), ("gaussian", True, _lib.KMVP_F32, 4, "cell_kernel"),
             ("gaussian", False, _lib.KMVP_F64, 3, "cell64_kernel"),
             ("absolute-exponential", True, _lib.KMVP_F64, 0, "lowd_kernel")]
    for kernel, norm, dtype, fast, kname in cases:
        npdt = np.float64 if dtype == _lib.KMVP_F64 else np.float32
        ctx = _lib.Context(0)
        try:
            ctx.comm_init(_lib.comm_unique_id(), 0, 1)
            ctx.set_option("fast_sqdists", fast)
            ctx.set_points(y.astype(npdt), None, dtype)
            ctx.set_signal(b.astype(npdt))
            ctx.run(kernel, norm)
            got = ctx.get_result(n, 1)
            assert ctx.last_kernel_name == kname
        finally:
            ctx.close()
        want = kmvp_oracle.product(kernel=kernel, source_points=y, source_signal=b, normalize_rows=norm)
        assert rel_err(got, want) <= (TOL64 if dtype == _lib.KMVP_F64 else TOL32), (kernel, kname)
    # bf16 matrix-core path, E > 1
    yd = np.random.RandomState(5).rand(300, 32) / np.sqrt(32)
    bd = np.random.RandomState(6).randn(300, 5)
    ctx = _lib.Context(0)
    try:
        ctx.comm_init(_lib.comm_unique_id(), 0, 1)
        ctx.set_points(yd.astype(np.float32), None, _lib.KMVP_BF16)
        ctx.set_signal(bd.astype(np.float32))
        ctx.run("absolute-exponential", True)
        got = ctx.get_result(300, 5)
    finally:
        ctx.close()
    want = kmvp_oracle.product(kernel="absolute-exponential", source_points=yd, source_signal=bd, normalize_rows=True)
    assert rel_err(got, want) <= TOL_BF16
    # several signal columns on the matrix cores (one block / three blocks of 32 columns), through the same exchange
    for kernel, fast, E, norm, kname in (("gaussian", 1, 5, True, "fastmm_kernel"), ("gaussian", 1, 70, False, "fastmm_kernel"),
                                         ("absolute-exponential", 2, 5, True, "cfastmm_kernel"),
                                         ("absolute-exponential", 2, 70, True, "cfastmm_kernel")):
        bm = np.random.RandomState(E).randn(n, E)
        ctx = _lib.Context(0)
        try:
            ctx.comm_init(_lib.comm_unique_id(), 0, 1)
            ctx.set_option("fast_sqdists", fast)
            ctx.set_points(y.astype(np.float32), None, _lib.KMVP_F32)
            ctx.set_signal(bm.astype(np.float32))
            ctx.run(kernel, norm)
            got = ctx.get_result(n, E)
            assert ctx.last_kernel_name == kname
        finally:
            ctx.close()
        want = kmvp_oracle.product(kernel=kernel, source_points=y, source_signal=bm, normalize_rows=norm)
        assert rel_err(got, want) <= TOL32, (kernel, kname, E)


def test_abi_error_behaviour():
    ctx = _lib.Context(0)
    try:
        with pytest.raises(_lib.KmvpError):  # no points yet
            ctx.run("gaussian", False)
        with pytest.raises(_lib.KmvpError):  # kmvp_fit wants the points too
            ctx.fit("gaussian")
        y = np.random.rand(10, 3).astype(np.float32)
        ctx.set_points(y, None, _lib.KMVP_F32)
        ctx.fit("inverse-distance")  # nothing to build: a no-op, not an error
        with pytest.raises(_lib.KmvpError):  # no signal yet
            ctx.run("gaussian", False)
        with pytest.raises(_lib.KmvpError):
            ctx.set_option("no-such-option", 1)
        with pytest.raises(_lib.KmvpError):
            ctx.set_option("targets_per_lane", 3)
        ctx.set_signal(np.ones((10, 1), dtype=np.float32))
        ctx.run("gaussian", False)
        assert ctx.get_result(10, 1).shape == (10, 1)
    finally:
        ctx.close()
    with pytest.raises(_lib.KmvpError):
        _lib.Context(10 ** 6)


def test_inputs_are_never_modified():
    y, b = kmvp_oracle.uniform_cube(500, 3)
    y0, b0 = y.copy(), b.copy()
    run_plugin(dict(kernel="inverse-distance", D=3), y, None, b, np.float32)
    assert np.array_equal(y, y0) and np.array_equal(b, b0)


# ---- the benchmark's full size: row subset against the oracle + size-independent properties

def test_config2_gaussian_1e6_row_subset_and_properties():
    n = 1_000_000
    y, b = kmvp_oracle.uniform_cube(n, 3)  # BASELINE config 2, datasets.py:256-266 recipe
    rows = np.random.RandomState(0).choice(n, size=1024, replace=False)
    algo = MI355XProduct(kernel="gaussian", dimension=3, precision="float32")
    try:
        algo.prepare_data(source_points=y, target_points=y, same_points=True)
        algo.prepare_query(source_signal=b)
        algo.query()
        a = algo.get_result()
        want = c_oracle.product(kernel="gaussian", source_points=y, source_signal=b, rows=rows)
        scale = np.max(np.abs(want))
        assert np.max(np.abs(a[rows] - want)) / scale <= TOL32
        # linearity: K(2b + c) = 2 Kb + Kc
        c = np.random.RandomState(1).randn(n, 1)
        algo.prepare_query(source_signal=c)
        algo.query()
        ac = algo.get_result()
        algo.prepare_query(source_signal=2 * b + c)
        algo.query()
        a2 = algo.get_result()
        assert np.max(np.abs(a2 - (2 * a + ac))) / np.max(np.abs(a2)) <= 4 * TOL32
        # symmetry of K when x == y:  <c, K b> = <b, K c>
        lhs, rhs = float(np.sum(c * a)), float(np.sum(b * ac))
        assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), np.linalg.norm(c) * np.linalg.norm(a) * 1e-2)
    finally:
        algo.done()


def test_config2_normalised_rows_are_convex_combinations():
    n = 200_000
    y, b = kmvp_oracle.uniform_cube(n, 3)
    for kernel in golden_cases.KERNELS:
        algo = MI355XProduct(kernel=kernel, dimension=3, normalize_rows=True, precision="float32")
        try:
            algo.prepare_data(source_points=y, target_points=y, same_points=True)
            algo.prepare_query(source_signal=b)
            algo.query()
            a = algo.get_result()
            assert a.min() >= b.min() - 1e-4 and a.max() <= b.max() + 1e-4
            algo.prepare_data(source_points=y, target_points=y, same_points=True, density_estimation=True)
            algo.prepare_query(source_signal=np.ones((n, 1)))
            algo.query()
            assert np.array_equal(algo.get_result(), np.ones((n, 1)))  # bruteforce.py:134-138
        finally:
            algo.done()


def test_inverse_distance_1e6_row_subset_with_sharded_offsets():
    """Config 4's kernel at 1e6, computed as 2 source shards on the one GPU."""
    n = 1_000_000
    y, b = kmvp_oracle.uniform_cube(n, 3)
    rows = np.random.RandomState(2).choice(n, size=512, replace=False)
    total = np.zeros((n, 1))
    for lo, hi in ((0, n // 2 + 7), (n // 2 + 7, n)):
        ctx = _lib.Context(0)
        try:
            ctx.set_option("same_points_global", 1)  # what the plugin sets for sharded same_points
            ctx.set_option("partial_shard", 1)
            ctx.set_points(np.ascontiguousarray(y[lo:hi], dtype=np.float32), y.astype(np.float32),
                           _lib.KMVP_F32, j_offset=lo, M_total=n)
            ctx.set_signal(np.ascontiguousarray(b[lo:hi], dtype=np.float32))
            ctx.run("inverse-distance", False)
            assert ctx.last_kernel_name == "cfast_kernel"
            total += ctx.get_result(n, 1)
        finally:
            ctx.close()
    assert np.isfinite(total).all()  # the diagonal was zeroed by GLOBAL index in both shards
    want = c_oracle.product(kernel="inverse-distance", source_points=y, source_signal=b, rows=rows)
    assert np.max(np.abs(total[rows] - want)) / np.max(np.abs(want)) <= 2 * TOL32


# ---- every BASELINE config at ITS OWN size ------------------------------------------

def test_config1_gaussian_1e4_float64_all_rows():
    """C1 (Gaussian, uniform-3D, N = M = 1e4, E = 1, float64): every row against the numpy oracle
    (bruteforce.py:25-58,130-153 restated), and the float32 run inside the north-star tolerance."""
    n = 10_000
    y, b = kmvp_oracle.uniform_cube(n, 3)  # seed 10003
    want = kmvp_oracle.product(kernel="gaussian", source_points=y, source_signal=b)
    got, extra = run_plugin(dict(kernel="gaussian", D=3), y, None, b, np.float64)
    assert got.shape == (n, 1) and rel_err(got, want) <= TOL64, rel_err(got, want)
    got32, _ = run_plugin(dict(kernel="gaussian", D=3), y, None, b, "float32")
    assert rel_err(got32, want) <= TOL32, rel_err(got32, want)


def test_config2_cell_kernel_on_all_rows_against_the_float64_difference_form():
    """C2 at full size, ALL 1e6 rows: the cell form (exp() range-reduced by grid cells, remainder polynomial on
    the matrix cores -- an approximating kernel) against the float64 difference form lowd_kernel<double>
    (golden-pinned to 1e-11 and, here, to the C oracle on 256 rows) on the same points, with the float32
    difference form beside it.  A mis-tiled boundary cell or a dropped tile shows up here; a 0.1 % row sample,
    or the linearity / symmetry checks, would let it through."""
    n = 1_000_000
    y, b = kmvp_oracle.uniform_cube(n, 3)
    truth_algo = MI355XProduct(kernel="gaussian", dimension=3, precision=np.float64, fast_sqdists=False)
    try:
        truth_algo.prepare_data(source_points=y, target_points=y, same_points=True)
        truth_algo.prepare_query(source_signal=b)
        truth_algo.query()
        assert truth_algo.device_kernel == "lowd_kernel"
        truth = truth_algo.get_result()
    finally:
        truth_algo.done()
    rows = np.random.RandomState(0).choice(n, size=256, replace=False)
    want = c_oracle.product(kernel="gaussian", source_points=y, source_signal=b, rows=rows)
    assert np.max(np.abs(truth[rows] - want)) / np.max(np.abs(want)) <= TOL64
    algo = MI355XProduct(kernel="gaussian", dimension=3, precision="float32")
    try:
        algo.prepare_data(source_points=y, target_points=y, same_points=True)
        algo.fit()
        algo.prepare_query(source_signal=b)
        algo.query()
        assert algo.device_kernel == "cellmm16_kernel"  # what auto picks at the headline shape (the 16x16x32 form of cellmm_kernel)
        cells = algo.get_result()
        algo.set_query_arguments(fast_sqdists=4)
        algo.query()
        assert algo.device_kernel == "cell_kernel"
        cells_valu = algo.get_result()
        algo.set_query_arguments(fast_sqdists=0)
        algo.query()
        assert algo.device_kernel == "lowd_kernel"
        diff = algo.get_result()
    finally:
        algo.done()
    scale = np.max(np.abs(truth))
    e_cell = np.max(np.abs(cells - truth)) / scale
    e_valu = np.max(np.abs(cells_valu - truth)) / scale
    e_diff = np.max(np.abs(diff - truth)) / scale
    print(f"config 2, all 1e6 rows vs float64: cellmm_kernel {e_cell:.2e}, cell_kernel {e_valu:.2e}, float32 difference form {e_diff:.2e}")
    assert e_cell <= 1e-6, e_cell   # measured 6-8e-7: every row, not a sample
    assert e_valu <= 1e-6, e_valu
    assert e_diff <= TOL32, e_diff


def test_cell_kernel_worst_case_placement_in_opposite_cell_corners():
    """The remainder polynomial of the cell form is truncated (exp(t) ~ 1 + t + t^2/2, |t| = |2 d.e| <= 0.016):
    its error is largest for a target in one corner of its cell and a source in the far corner of a cell
    diagonal to it (d and e parallel, both at the full half-diagonal).  This cloud is built to be exactly that:
    every point sits within 2 % of a cell side of a corner of its cell, half of the cells' points in the
    (-,-,-) corner and half in the (+,+,+) corner, 65 536 points, so every target sees its worst-placed sources
    at EVERY distance.  Checked against the float64 oracle on all rows; the difference form on the same data
    is the yardstick."""
    rs = np.random.RandomState(77)
    g = 8                       # 8 x 8 x 8 cells over [0, 0.8]^3: side 0.1 <= sqrt(0.032 / 3) = 0.1033
    n_cell = 128                # 65 536 points >= 32768, four full tiles per cell
    cells = np.stack(np.meshgrid(np.arange(g), np.arange(g), np.arange(g), indexing="ij"), axis=-1).reshape(-1, 3)
    h = 0.1
    pts = []
    for c in cells:
        lo_corner = c * h + rs.rand(n_cell // 2, 3) * 0.02 * h
        hi_corner = (c + 1) * h - rs.rand(n_cell // 2, 3) * 0.02 * h
        pts.append(lo_corner)
        pts.append(hi_corner)
    y = np.concatenate(pts)
    # pin the bounding box to [0, 0.8]^3 so that the library's box-fitted grid IS this grid
    y[0] = 0.0
    y[-1] = g * h
    y = y[rs.permutation(len(y))]
    n = len(y)
    assert n >= 32768
    b = np.abs(rs.randn(n, 1)) + 0.5  # one sign: truncation errors (all of one sign for parallel d, e) cannot cancel
    want = c_oracle.product(kernel="gaussian", source_points=y, source_signal=b, rows=np.arange(n))  # float64, all rows
    got, extra = run_plugin(dict(kernel="gaussian", D=3), y, None, b, "float32", fast_sqdists="cells")
    assert extra["device_kernel"] == "cellmm_kernel"
    got_valu, extra = run_plugin(dict(kernel="gaussian", D=3), y, None, b, "float32", fast_sqdists="cells-valu")
    assert extra["device_kernel"] == "cell_kernel"
    assert rel_err(got_valu, want) <= 1.5e-6, rel_err(got_valu, want)
    ref, extra0 = run_plugin(dict(kernel="gaussian", D=3), y, None, b, "float32", fast_sqdists=False)
    assert extra0["device_kernel"] == "lowd_kernel"
    e_cell, e_diff = rel_err(got, want), rel_err(ref, want)
    assert e_cell <= TOL32, (e_cell, e_diff)
    # the design bound (kmvp_cell.hpp): <= 6.8e-7 truncation + ~5e-7 of bf16 rounding for the worst placed pair
    assert e_cell <= 1.5e-6, (e_cell, e_diff)
    # whatever the plugin picks by itself on this cloud stays inside the tolerance as well
    auto, extra_auto = run_plugin(dict(kernel="gaussian", D=3), y, None, b, "float32")
    assert rel_err(auto, want) <= TOL32, extra_auto
    print(f"worst-case placement: cellmm_kernel {e_cell:.2e}, cell_kernel {rel_err(got_valu, want):.2e}, difference form {e_diff:.2e}, "
          f"auto = {extra_auto['device_kernel']}")


def test_cellmm_kernel_both_mfma_shapes():
    """The float32 cell form on both matrix-core shapes -- cellmm_kernel (32x32x16 f16) and cellmm16_kernel (16x16x32: two
    groups of 16 sources per MFMA, 16 row buckets; the default at the headline size) -- for every tile count, plain /
    density / normalised, a uniform and a clustered cloud with empty and overfull cells, targets != sources: each against
    the float64 oracle at the float32 tolerance and against the other shape (same algebra, other summation order)."""
    rs = np.random.RandomState(16)
    clouds = {"uniform": rs.rand(60000, 3), "clustered": np.concatenate([rs.rand(30000, 3) * 0.2 + 0.4, rs.rand(30000, 3)])}
    for name, y in clouds.items():
        b = rs.randn(len(y), 1)
        x = rs.rand(40000, 3)
        for targets, norm, dens in ((None, False, False), (x, False, False), (None, True, False), (None, False, True)):
            want = kmvp_oracle.product(kernel="gaussian", source_points=y, target_points=targets, source_signal=None if dens else b,
                                       normalize_rows=norm, density_estimation=dens, rows=np.arange(0, 40000, 97))
            for tiles in (1, 2, 4, 8):
                outs = []
                for shape in (0, 1):
                    algo = MI355XProduct(kernel="gaussian", dimension=3, normalize_rows=norm, precision="float32", fast_sqdists="cells",
                                         fast_tiles=tiles)
                    try:
                        algo.prepare_data(source_points=y, target_points=y if targets is None else targets,
                                          same_points=targets is None, density_estimation=dens)
                        algo.set_query_arguments(cellmm_shape=shape)
                        algo.fit()
                        algo.prepare_query(source_signal=b)
                        algo.query()
                        outs.append(algo.get_result())
                        assert algo.device_kernel == ("cellmm16_kernel" if shape else "cellmm_kernel")
                    finally:
                        algo.done()
                    assert rel_err(outs[-1][::97][: len(want)], want) <= TOL32, (name, tiles, shape, norm, dens)
                assert rel_err(outs[1], outs[0]) <= 5e-7, (name, tiles, rel_err(outs[1], outs[0]))


def test_config4_inverse_distance_one_of_eight_shards_at_full_size():
    """C4 (inverse-distance, uniform-3D, N = M = 1e7, E = 1, float32, sources sharded over 8 GPUs): what ONE of
    the 8 ranks computes -- all 1e7 targets x its 1.25e6 sources, with the zero rule of bruteforce.py:8-15 on
    the GLOBAL flat index (j_offset / M_total) -- at the config's own size.  All rows finite (the diagonal
    falls inside rank 3's slice and must be zeroed there); 512 rows against the C oracle's shard sums."""
    n = 10_000_000
    y, b = kmvp_oracle.uniform_cube(n, 3)  # seed 10000003
    rank, world = 3, 8
    lo, hi = n * rank // world, n * (rank + 1) // world
    y32 = y.astype(np.float32)
    ctx = _lib.Context(0)
    try:
        ctx.set_option("same_points_global", 1)  # what the plugin sets for sharded same_points
        ctx.set_option("partial_shard", 1)       # one rank's partial sums, on purpose (no communicator here)
        ctx.set_points(np.ascontiguousarray(y32[lo:hi]), y32, _lib.KMVP_F32, j_offset=lo, M_total=n)
        ctx.set_signal(np.ascontiguousarray(b[lo:hi], dtype=np.float32))
        ctx.run("inverse-distance", False)
        kname, ms, dev_gb = ctx.last_kernel_name, ctx.last_kernel_ms, ctx.device_bytes / 1e9
        part = ctx.get_result(n, 1)
    finally:
        ctx.close()
    assert kname == "cfast_kernel", kname
    assert np.isfinite(part).all()
    rows = np.concatenate([np.random.RandomState(4).choice(n, size=448, replace=False),
                           np.arange(lo, lo + 32), np.arange(hi - 32, hi)])  # incl. rows whose zero column is in the slice
    want, _ = c_oracle.product(kernel="inverse-distance", source_points=y[lo:hi], target_points=y, source_signal=b[lo:hi],
                               rows=rows, j_offset=lo, M_total=n, raw_sums=True)
    err = np.max(np.abs(part[rows] - want)) / np.max(np.abs(want))
    assert err <= 2 * TOL32, err
    print(f"config 4 shard: {kname} {ms:.1f} ms, {n * (hi - lo) / ms / 1e9:.2f}e12 pairs/s, {dev_gb:.2f} GB, rel err {err:.2e}")


def test_config4_all_eight_shards_sum_to_the_whole_product_at_full_size():
    """C4's whole arithmetic on the one GPU a test box has: the eight ranks' source slices (sharding.shard_range, the
    plugin's own split) run one after the other -- same context, `partial_shard` = 1, j_offset / M_total of each rank --
    and their float64 partial sums are added on the host, which is what the single RCCL all-reduce of SURVEY 8e does.
    Checks: every row of the sum finite (each target's zero column, bruteforce.py:8-15, lies in exactly one slice and
    must be zeroed there and nowhere else); 512 rows -- incl. the rows on both sides of every slice edge, whose zero
    column is the first / last source of a slice -- against the float64 C oracle on ALL 1e7 sources."""
    from kernel_matrix_benchmarks_amd import sharding

    n, world = 10_000_000, 8
    y, b = kmvp_oracle.uniform_cube(n, 3)  # seed 10000003
    y32 = y.astype(np.float32)
    b32 = b.astype(np.float32)
    total = np.zeros((n, 1))
    edges, ms_all, names = [], [], set()
    ctx = _lib.Context(0)
    try:
        ctx.set_option("same_points_global", 1)
        ctx.set_option("partial_shard", 1)
        for rank in range(world):
            lo, hi = sharding.shard_range(n, rank, world)
            edges += [lo, min(lo + 1, n - 1), hi - 2, hi - 1]
            ctx.set_points(np.ascontiguousarray(y32[lo:hi]), y32, _lib.KMVP_F32, j_offset=lo, M_total=n)
            ctx.set_signal(np.ascontiguousarray(b32[lo:hi]))
            ctx.run("inverse-distance", False)
            names.add(ctx.last_kernel_name)
            ms_all.append(ctx.last_kernel_ms)
            part = ctx.get_result(n, 1)
            assert np.isfinite(part).all(), f"shard {rank}: non-finite partial sums"
            total += part
    finally:
        ctx.close()
    assert names == {"cfast_kernel"}, names
    rows = np.unique(np.concatenate([np.random.RandomState(8).choice(n, size=512 - len(edges), replace=False), np.array(edges)]))
    want = c_oracle.product(kernel="inverse-distance", source_points=y, source_signal=b, rows=rows)
    err = np.max(np.abs(total[rows] - want)) / np.max(np.abs(want))
    assert err <= 2 * TOL32, err
    print(f"config 4, 8 shards on one GPU: {sum(ms_all):.0f} ms of pair loops ({n * float(n) / sum(ms_all) / 1e9:.2f}e12 pairs/s), "
          f"{len(rows)} rows rel err {err:.2e}")


def test_config5_gaussian_cg_solver_at_1e5_float64():
    """C5 exactly (Gaussian solver K b = a, N = M = 1e5, D = 3, float64, CG with the HIP matvec as operator,
    residual < 1e-6): uniform_cube(100000, 3) (seed 100003), a := K b from the float64 product,
    MI355XSolver(rtol=1e-6).  Judged on the residual (the reference's lstsq, bruteforce.py:193-207,
    cannot run at 80 GB): the solver's own true residual, and the oracle's on 256 rows."""
    n = 100_000
    y, b = kmvp_oracle.uniform_cube(n, 3)
    prod = MI355XProduct(kernel="gaussian", dimension=3, precision=np.float64)
    try:
        prod.prepare_data(source_points=y, target_points=y, same_points=True)
        prod.fit()
        prod.prepare_query(source_signal=b)
        prod.query()
        a = prod.get_result()
        assert prod.device_kernel == "cell64_kernel"
    finally:
        prod.done()
    rows = np.random.RandomState(3).choice(n, size=256, replace=False)
    a_rows = c_oracle.product(kernel="gaussian", source_points=y, source_signal=b, rows=rows)
    assert np.max(np.abs(a[rows] - a_rows)) / np.max(np.abs(a_rows)) <= TOL64
    algo = MI355XSolver(kernel="gaussian", dimension=3, precision=np.float64, rtol=1e-6, maxit=2000)
    try:
        algo.prepare_data(source_points=y)
        algo.fit()
        algo.prepare_query(target_signal=a)
        algo.query()
        sol = algo.get_result()
        info = algo.get_additional()
    finally:
        algo.done()
    assert info["cg_converged"], info
    assert info["device_kernel"] == "cell64_kernel", info
    assert info["cg_relative_residual"] <= 1.5e-6, info  # the TRUE residual (kmvp.h), 1.5x slack over the recurrence's
    assert sol.shape == (n, 1) and np.isfinite(sol).all()
    Kb = c_oracle.product(kernel="gaussian", source_points=y, source_signal=sol, rows=rows)
    res_rows = np.linalg.norm(Kb - a[rows]) / np.linalg.norm(a[rows])
    assert res_rows <= 1e-5, (res_rows, info)
    print(f"config 5: {info['cg_iterations']} iterations, residual {info['cg_relative_residual']:.2e}, rows {res_rows:.2e}")


def test_mixed_precision_refinement_reaches_the_float64_residual():
    """MI355XSolver(refine="float32") (an extension; the reference has one dense lstsq): float64 residuals, float32 CG
    corrections on the matrix-core operator.  Judged like the plain float64 solve: true residual <= 1.5 rtol, checked
    on 256 oracle rows; an inner tolerance the float32 operator cannot reach must end as NOT converged with the last
    good iterate, never with garbage."""
    n = 100_000  # config 5's cloud: dense enough for the float32 cell form (cellmm_kernel) as inner operator
    y, b = kmvp_oracle.uniform_cube(n, 3)
    a = run_plugin(dict(kernel="gaussian", D=3), y, None, b, np.float64)[0]
    rows = np.random.RandomState(3).choice(n, size=256, replace=False)
    out = {}
    for inner in (1e-3, 1e-6):
        sol = MI355XSolver(kernel="gaussian", dimension=3, precision=np.float64, rtol=1e-6, maxit=3000, refine="float32",
                           inner_rtol=inner)
        try:
            sol.prepare_data(source_points=y)
            sol.fit()
            sol.prepare_query(target_signal=a)
            sol.query()
            out[inner] = (sol.get_result(), sol.get_additional())
        finally:
            sol.done()
    x, info = out[1e-3]
    assert info["cg_converged"] and info["cg_relative_residual"] <= 1.5e-6, info
    assert info["inner_device_kernel"] == "cellmm_kernel" and info["refinement_steps"] >= 2, info
    assert info["refinement_stop_reason"] == "tolerance" and np.isfinite(info["refinement_last_inner_residual"]), info
    Kx = c_oracle.product(kernel="gaussian", source_points=y, source_signal=x, rows=rows)
    assert np.linalg.norm(Kx - a[rows]) / np.linalg.norm(a[rows]) <= 5e-6
    x6, info6 = out[1e-6]
    assert np.isfinite(x6).all() and np.isfinite(info6["cg_relative_residual"])
    if not info6["cg_converged"]:
        assert info6["cg_relative_residual"] <= 1.0  # the last good iterate (at worst x = 0), not a diverged one
        assert info6["refinement_stop_reason"] in ("stagnation", "maxit", "outer-limit", "non-finite"), info6
    else:
        assert info6["refinement_stop_reason"] == "tolerance", info6
    with pytest.raises(NotImplementedError):
        MI355XSolver(kernel="inverse-distance", dimension=3, precision=np.float64, refine="float32")
    with pytest.raises(NotImplementedError):
        MI355XSolver(kernel="gaussian", dimension=3, precision="float32", refine="float32")


def test_solver_verdict_is_the_true_residual_and_never_nan():
    """KMVP_OK only for a finite TRUE residual <= 1.5 rtol.  A NaN right-hand side, or an operator
    with inf entries (inverse-distance with a duplicated point under MINRES), is not a success."""
    n = 600
    y, _ = kmvp_oracle.uniform_cube(n, 3)
    a = np.random.RandomState(1).randn(n, 1)
    bad = a.copy()
    bad[17, 0] = np.nan
    for kernel in ("gaussian", "inverse-distance"):
        ctx = _lib.Context(0)
        try:
            ctx.set_points(y, None, _lib.KMVP_F64)
            sol, iters, resid, ok = ctx.cg_solve(kernel, bad, 1e-8, 200)
        finally:
            ctx.close()
        assert not ok and not np.isfinite(resid), (kernel, iters, resid, ok)
    dup = y.copy()
    dup[5] = dup[400]  # coincident off-diagonal points: 1/sqrt(0) = inf in the matrix (bruteforce.py:8-15)
    ctx = _lib.Context(0)
    try:
        ctx.set_points(dup, None, _lib.KMVP_F64)
        sol, iters, resid, ok = ctx.cg_solve("inverse-distance", a, 1e-8, 200)
    finally:
        ctx.close()
    assert not ok, (iters, resid)
    # a float32 operator on an ill-conditioned Gaussian system: whatever is returned as converged has a TRUE
    # residual within 1.5 rtol (checked against the float64 oracle, not against the solver's own number)
    n2 = 3000
    y2, b2 = kmvp_oracle.uniform_cube(n2, 3)
    a2 = kmvp_oracle.product(kernel="gaussian", source_points=y2, source_signal=b2)
    ctx = _lib.Context(0)
    try:
        ctx.set_points(y2.astype(np.float32), None, _lib.KMVP_F32)
        sol, iters, resid, ok = ctx.cg_solve("gaussian", a2.astype(np.float32), 1e-4, 3000)
    finally:
        ctx.close()
    true = kmvp_oracle.relative_residual(kernel="gaussian", source_points=y2, solution=sol, target_signal=a2)
    assert ok == (resid <= 1.5e-4)
    if ok:
        assert true <= 3e-4, (true, resid)  # float32 operator vs float64 oracle: a factor of 2 between the two views


def test_runner_drives_the_headline_dataset_under_its_reference_name(tmp_path):
    """the large synthetic config registered under the reference's naming pattern
    (algos.yaml:38) and driven through the runner: product-cube-D3-E1-M1000000-N1000000-gaussian.
    The truth is the float64 HIP product (datasets.ground_truth), itself pinned at 1e6 by the oracle rows here."""
    from kernel_matrix_benchmarks_amd import datasets, metrics, runner

    name = "product-cube-D3-E1-M1000000-N1000000-gaussian"
    stored = runner.run_dataset(name, hardware="GPU", runs=2, data_root=str(tmp_path / "data"),
                                results_root=str(tmp_path / "results"), verbose=False)
    by_name = {attrs["name"]: (attrs, result) for _, attrs, result in stored}
    assert sorted(by_name) == ["MI355XProduct(float16)", "MI355XProduct(float32)", "MI355XProduct(float64)"]
    f, D = datasets.get_dataset(name, root=str(tmp_path / "data"))
    try:
        y = np.asarray(f["source_points"][:])
        b = np.asarray(f["source_signal"][:])
        truth = np.asarray(f["target_signal"][:])
        assert bool(f.attrs["same_points"]) and f.attrs["kernel"] == "gaussian"
    finally:
        f.close()
    yo, bo = kmvp_oracle.uniform_cube(1_000_000, 3)
    assert np.array_equal(y, yo) and np.array_equal(b, bo)
    rows = np.random.RandomState(5).choice(1_000_000, size=256, replace=False)
    want = c_oracle.product(kernel="gaussian", source_points=y, source_signal=b, rows=rows)
    assert np.max(np.abs(truth[rows] - want)) / np.max(np.abs(want)) <= TOL64
    a32, r32 = by_name["MI355XProduct(float32)"]
    a64, r64 = by_name["MI355XProduct(float64)"]
    assert a32["device_kernel"] == "cellmm16_kernel" and a64["device_kernel"] == "cell64_kernel"
    assert metrics.relative_max_error(r32, truth) <= TOL32 and metrics.relative_max_error(r64, truth) <= TOL64
    assert a32["query_time"] < 0.2 and a32["build_time"] < 0.2 and a32["run_count"] == 2


# ---- solver

def test_cg_solver_reaches_the_residual(expected):
    for case in golden_cases.solver_cases():
        y, _ = golden_cases.make_solver_inputs(case)
        a = expected[f"{case['name']}/a"]
        algo = MI355XSolver(kernel=case["kernel"], dimension=3, precision=np.float64, rtol=1e-8, maxit=5000)
        try:
            algo.prepare_data(source_points=y)
            algo.fit()
            algo.prepare_query(target_signal=a)
            algo.query()
            sol = algo.get_result()
            info = algo.get_additional()
        finally:
            algo.done()
        assert sol.shape == a.shape and sol.dtype == np.float64
        res = kmvp_oracle.relative_residual(kernel=case["kernel"], source_points=y, solution=sol, target_signal=a)
        assert info["cg_converged"], (case["name"], info)
        assert res <= 2e-8, (case["name"], res, info)
        assert abs(res - info["cg_relative_residual"]) <= 1e-9


def test_minres_solver_on_the_references_sphere_datasets():
    """solver-sphere-*-inverse-distance (datasets.py:393-399): symmetric indefinite matrix -> MINRES."""
    for n in (500, 2000):
        y = kmvp_oracle.uniform_sphere_points(n)
        b_true = np.random.RandomState(n).randn(n, 2)
        a = kmvp_oracle.product(kernel="inverse-distance", source_points=y, source_signal=b_true)
        algo = MI355XSolver(kernel="inverse-distance", dimension=3, precision=np.float64, rtol=1e-8, maxit=20000)
        try:
            algo.prepare_data(source_points=y)
            algo.fit()
            algo.prepare_query(target_signal=a)
            algo.query()
            sol = algo.get_result()
            info = algo.get_additional()
        finally:
            algo.done()
        res = kmvp_oracle.relative_residual(kernel="inverse-distance", source_points=y, solution=sol, target_signal=a)
        assert info["cg_converged"] and res <= 2e-8, (n, res, info)
        # this matrix is well conditioned enough for the iterate to match the dense lstsq answer
        ref = kmvp_oracle.solve(kernel="inverse-distance", source_points=y, target_signal=a)
        assert np.max(np.abs(sol - ref)) <= 1e-5 * np.max(np.abs(ref)), np.max(np.abs(sol - ref))


def test_cg_solver_config5_shape_small():
    """Config 5 recipe (gaussian, D=3, fp64, a := K b) at n = 20000: residual < 1e-6."""
    n = 20000
    y, b = kmvp_oracle.uniform_cube(n, 3)
    prod = MI355XProduct(kernel="gaussian", dimension=3, precision=np.float64)
    try:
        prod.prepare_data(source_points=y, target_points=y, same_points=True)
        prod.prepare_query(source_signal=b)
        prod.query()
        a = prod.get_result()
    finally:
        prod.done()
    algo = MI355XSolver(kernel="gaussian", dimension=3, precision=np.float64, rtol=1e-6, maxit=2000)
    try:
        algo.prepare_data(source_points=y)
        algo.prepare_query(target_signal=a)
        algo.query()
        sol = algo.get_result()
        info = algo.get_additional()
    finally:
        algo.done()
    rows = np.random.RandomState(3).choice(n, size=256, replace=False)
    Kb = c_oracle.product(kernel="gaussian", source_points=y, source_signal=sol, rows=rows)
    assert info["cg_relative_residual"] <= 1e-6, info
    assert np.linalg.norm(Kb - a[rows]) / np.linalg.norm(a[rows]) <= 5e-6


def test_long_cg_graph_replay_equals_plain_launches():
    """After 512 iterations the CG burst is replayed as a hipGraph: same iterates as launch by launch."""
    import os

    n = 1200
    y, b = kmvp_oracle.uniform_cube(n, 3)
    y = y * 4.0
    a = kmvp_oracle.product(kernel="gaussian", source_points=y, source_signal=b)
    outs = []
    for no_graph in (False, True):
        if no_graph:
            os.environ["KMVP_NO_GRAPH"] = "1"
        else:
            os.environ.pop("KMVP_NO_GRAPH", None)
        ctx = _lib.Context(0)
        try:
            ctx.set_points(y, None, _lib.KMVP_F64)
            sol, iters, resid, ok = ctx.cg_solve("gaussian", a, 1e-13, 1000)  # unreachable tolerance: runs to maxit
        finally:
            ctx.close()
            os.environ.pop("KMVP_NO_GRAPH", None)
        assert iters == 1000
        outs.append(sol)
    assert np.array_equal(outs[0], outs[1])


def test_long_cg_graph_replay_on_the_cell_operator():
    """The same with the float64 cell operator (N >= 32768): its per-iteration launches (signal re-pack, pair
    loop, segment sums, gather) are captured and replayed like any other product."""
    import os

    n = 33000
    rs = np.random.RandomState(9)
    y = rs.rand(n, 3)
    a = rs.randn(n, 1)
    outs = []
    for no_graph in (False, True):
        if no_graph:
            os.environ["KMVP_NO_GRAPH"] = "1"
        else:
            os.environ.pop("KMVP_NO_GRAPH", None)
        ctx = _lib.Context(0)
        try:
            ctx.set_option("fast_sqdists", 3)  # 33000 points sit at the padding limit of the auto rule
            ctx.set_points(y, None, _lib.KMVP_F64)
            sol, iters, resid, ok = ctx.cg_solve("gaussian", a, 1e-15, 560)  # unreachable tolerance: runs to maxit
            assert ctx.last_kernel_name == "cell64_kernel"
        finally:
            ctx.close()
            os.environ.pop("KMVP_NO_GRAPH", None)
        assert iters == 560
        outs.append(sol)
    assert np.array_equal(outs[0], outs[1])


def test_solver_in_the_sharded_shape_on_one_gpu():
    """The multi-GPU solver hands every rank ALL points as targets and a slice as
    sources, Krylov vectors replicated.  With one rank the slice is everything: same operator,
    same iterates as the x == y form; a partial slice without a communicator must be refused."""
    n = 3000
    y, b = kmvp_oracle.uniform_cube(n, 3)
    a = kmvp_oracle.product(kernel="absolute-exponential", source_points=y, source_signal=b)
    sols = []
    for sharded_shape in (False, True):
        ctx = _lib.Context(0)
        try:
            if sharded_shape:
                ctx.comm_init(_lib.comm_unique_id(), 0, 1)
                ctx.set_option("same_points_global", 1)
                ctx.set_points(y, y, _lib.KMVP_F64, j_offset=0, M_total=n)
            else:
                ctx.set_points(y, None, _lib.KMVP_F64)
            sol, iters, resid, ok = ctx.cg_solve("absolute-exponential", a, 1e-9, 5000)
        finally:
            ctx.close()
        assert ok and resid <= 2e-9, (sharded_shape, iters, resid)
        sols.append((sol, iters))
    assert sols[0][1] == sols[1][1]
    assert np.max(np.abs(sols[0][0] - sols[1][0])) <= 1e-9 * np.max(np.abs(sols[0][0]))
    assert np.max(np.abs(sols[0][0] - b)) <= 1e-5 * np.max(np.abs(b))

    ctx = _lib.Context(0)
    try:
        ctx.set_option("same_points_global", 1)
        ctx.set_points(np.ascontiguousarray(y[: n // 2]), y, _lib.KMVP_F64, j_offset=0, M_total=n)
        with pytest.raises(_lib.KmvpError):
            ctx.cg_solve("absolute-exponential", a, 1e-9, 10)
    finally:
        ctx.close()


# ---- the harness end to end: dataset file -> registry -> runner protocol -> result files

def test_runner_end_to_end_on_gpu(tmp_path):
    from kernel_matrix_benchmarks_amd import datasets, metrics, results, runner, storage

    name = "product-cube-D3-E1-M3000-N3000-inverse-distance"
    data_root, results_root = str(tmp_path / "data"), str(tmp_path / "results")
    stored = runner.run_dataset(name, hardware="GPU", runs=2, data_root=data_root,
                                results_root=results_root, verbose=False)
    assert len(stored) == 3  # float16, float32 and float64 run-group entries of algos.yaml
    # the generated dataset follows the reference's recipe and its truth is the oracle's answer
    f, D = datasets.get_dataset(name, root=data_root)
    try:
        y = np.asarray(f["source_points"][:])
        b = np.asarray(f["source_signal"][:])
        truth = np.asarray(f["target_signal"][:])
        assert D == 3 and bool(f.attrs["same_points"]) and not bool(f.attrs["density_estimation"])
        assert f.attrs["kernel"] == "inverse-distance" and f.attrs["task"] == "product"
    finally:
        f.close()
    yo, bo = kmvp_oracle.uniform_cube(3000, 3)
    assert np.array_equal(y, yo) and np.array_equal(b, bo)
    want = kmvp_oracle.product(kernel="inverse-distance", source_points=y, source_signal=b)
    assert rel_err(truth, want) <= TOL64
    for fn, attrs, result in stored:
        assert fn.endswith(storage.extension()) and attrs["algo"] == "mi355x-product"
        assert attrs["build_time"] < 0.05 and attrs["query_time"] > 0 and attrs["n_gpus"] == 1
    by_name = {attrs["name"]: (fn, result) for fn, attrs, result in stored}
    err64 = metrics.relative_max_error(by_name["MI355XProduct(float64)"][1], truth)
    err32 = metrics.relative_max_error(by_name["MI355XProduct(float32)"][1], truth)
    assert err64 <= TOL64 and err32 <= 2 * TOL32
    loaded = list(results.load_all_results(name, root=results_root))
    assert sorted(p["name"] for p, _ in loaded) == sorted(by_name)


def test_runner_solver_dataset_on_gpu(tmp_path):
    from kernel_matrix_benchmarks_amd import runner

    for name in ("solver-cube-D3-E1-M400-N400-absolute-exponential",
                 "solver-sphere-D3-E1-M400-N400-inverse-distance"):  # the reference's dataset family
        stored = runner.run_dataset(name, hardware="GPU", runs=1, data_root=str(tmp_path / "data"),
                                    results_root=str(tmp_path / "results"), verbose=False)
        assert len(stored) == 2
        for fn, attrs, result in stored:
            assert attrs["algo"] == "mi355x-solver" and result.shape == (400, 1)
            assert attrs["cg_converged"], attrs
            # the solvers stop on the recurrence residual; the TRUE residual they report may exceed rtol
            # by rounding (accepted up to 1.5 rtol, kmvp_solvers.hip)
            assert attrs["cg_relative_residual"] <= (1.5e-6 if "float64" in attrs["name"] else 1.5e-4)


def test_rccl_binds_to_the_hip_runtime_in_use():
    """A process can hold two ROCm stacks (system + the copy bundled with PyTorch).  libkmvp.so
    must dlopen the RCCL that belongs to the HIP runtime it is itself bound to, in both import
    orders; bench.py uses the libkmvp-first order (tools/rccl_stack_check.py)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_stack_check.py")], cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert "kmvp_first exit 0" in out.stdout and "torch_first exit 0" in out.stdout, out.stdout + out.stderr


def test_randomised_parity_sweep():
    """tools/fuzz_parity.py: 250 random combinations of kernel, D (1 .. 70), E (1 .. 65), N, M (1 .. 40000), same_points,
    normalize_rows, density_estimation, precision and squared-distance form through the plugin, each against the float64
    numpy oracle on the inputs as the working precision sees them (3000 more cases over other seeds were run by hand:
    profiles/r02_fuzz_parity.txt)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(root, "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    seen, failures = fuzz.sweep(250, 2024, verbose=False)
    assert not failures, failures
    assert {"lowd_kernel", "lowd_mid_kernel", "fast_kernel", "fastmm_kernel", "cfast_kernel", "cfastmm_kernel"} <= set(seen), seen


def test_randomised_solver_sweep():
    """tools/fuzz_solver.py: 60 random solves (kernel, n 40 .. 40000, D, one or three right-hand sides, float32 / float64,
    rtol, mixed-precision refinement; at most 600 iterations).  Whatever is reported as converged must have a true float64
    residual within 2 rtol (oracle), and ill-conditioned systems that do not get there must say so."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_solver", os.path.join(root, "tools", "fuzz_solver.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    stats, failures = fuzz.sweep(60, 5, verbose=False)
    assert not failures, failures
    assert stats["converged"] >= 10 and stats["not_converged"] >= 1, stats


@pytest.mark.parametrize("kernel,side,want_kernel", [("absolute-exponential", 6.0, "cfastmm_kernel"),
                                                     ("gaussian", 2.4, "fastmm_kernel"), ("gaussian", 9.0, "cfastmm_kernel")])
def test_multi_column_kernels_as_solver_operators(kernel, side, want_kernel):
    """K B = A with five right-hand sides in float32: the CG iterate is an (M, 5) signal, so every operator application
    inside the device-resident iteration (hipGraph replay for long solves) is a multi-column product -- fastmm_kernel inside
    the radius rule, cfastmm_kernel outside it and for exp(-r).  The verdict is checked against the float64 oracle."""
    n, E, rtol = 3000, 5, 1e-3
    rs = np.random.RandomState(int(side * 10))
    y = (rs.rand(n, 3) * side).astype(np.float32).astype(np.float64)
    a = rs.randn(n, E).astype(np.float32).astype(np.float64)
    algo = MI355XSolver(kernel=kernel, dimension=3, precision="float32", rtol=rtol, maxit=2000)
    try:
        algo.prepare_data(source_points=y)
        algo.fit()
        algo.prepare_query(target_signal=a)
        algo.query()
        b = algo.get_result()
        info = algo.get_additional()
    finally:
        algo.done()
    assert info["device_kernel"] == want_kernel, info
    Kb = kmvp_oracle.product(kernel=kernel, source_points=y, source_signal=b)
    res = float(np.max(np.linalg.norm(Kb - a, axis=0) / np.linalg.norm(a, axis=0)))
    if info["cg_converged"]:
        assert res <= 2 * rtol, (res, info)
    else:  # an ill-conditioned draw: it must have said so, with a residual that is the true one
        assert info["cg_relative_residual"] > rtol and abs(res - info["cg_relative_residual"]) <= 0.5 * res, (res, info)


@pytest.mark.parametrize("config,points,tol", [("2", 60000, 1e-5), ("2shard", 200000, 1e-5), ("3", 4096, 1e-2), ("softmax", 4096, 1e-2),
                                               ("attn", 20000, 1e-5), ("4shard", 200000, 2e-4), ("5", 6000, None)])
def test_bench_configs_at_reduced_size(config, points, tol):
    """`python bench.py --config C --points n` for every config the default run reports (headline and `other_configs`): one
    JSON line with the contract's keys, a roofline object, the error leg inside the working precision's tolerance (solver: the
    residual on the oracle rows).  The default-size numbers are the driver's; this keeps every config's code path alive."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", config, "--points", str(points), "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic"
    r = d["roofline"]
    assert r["bound"] in ("mfma", "valu", "hbm") and r["achieved"] > 0 and 0 < r["frac"] < 1.0 and r["kernel"].endswith("_kernel")
    if tol is not None:
        assert d["max_rel_err"] <= tol, d["max_rel_err"]
    else:
        assert d["solver"]["converged"] and d["solver"]["residual_rows"] <= 2e-6, d["solver"]
