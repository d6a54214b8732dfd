"""Randomised parity sweep: random shapes / kernels / query() branches / precisions / squared-distance forms through the
plugin, every result against the float64 numpy oracle.  Prints one line per failure and a summary; exit code 1 on any.
usage: python tools/fuzz_parity.py [cases=300] [seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct  # noqa: E402
import kmvp_oracle  # noqa: E402  (checker only)

KERNELS = ("gaussian", "absolute-exponential", "inverse-distance")
FORMS = (None, None, False, True, "centred", "cells", "cells-valu")


def one_case(rs):
    kernel = KERNELS[rs.randint(3)]
    D = int(rs.choice([1, 2, 3, 3, 3, 4, 5, 8, 9, 16, 40, 70]))
    E = int(rs.choice([1, 1, 1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 32, 33, 65]))
    N = int(rs.choice([1, 7, 31, 32, 33, 100, 257, 1000, 4097, 40000]))
    M = int(rs.choice([1, 5, 31, 32, 33, 64, 127, 129, 1000, 3001, 40000]))
    same = bool(rs.rand() < 0.3)
    if same:
        N = M
    norm = bool(rs.rand() < 0.4)
    dens = bool(rs.rand() < 0.15)
    precision = ["float32", "float32", "float64", "float16"][rs.randint(4)]
    form = FORMS[rs.randint(len(FORMS))]
    tiles = int(rs.choice([0, 0, 1, 2, 4, 8]))
    spread = float(rs.choice([0.3, 1.0, 1.0, 3.0]))
    # targets displaced from the sources by this many kernel lengths (targets != sources only): every kernel value of a
    # row is then tiny, and each ROW is held to its own mass sum_j k |b_j| (the per-target shift of the f16-split kernels)
    offset = 0.0 if same else float(rs.choice([0.0, 0.0, 0.0, 2.0, 4.0, 7.0]))
    if rs.rand() < 0.12:  # exp<x, y>: the online-max kernel (float32, D <= 64) or the Gaussian identity
        kernel, form, tiles, dens = "exp-dot", None, 0, False
        precision = ["float32", "float32", "float64", "bfloat16"][rs.randint(4)]
    return dict(kernel=kernel, D=D, E=E, N=N, M=M, same=same, norm=norm, dens=dens, precision=precision, form=form,
                tiles=tiles, spread=spread, offset=offset)


def run_case(c, rs):
    D, E, N, M = c["D"], c["E"], c["N"], c["M"]
    y = rs.rand(M, D) * c["spread"] / np.sqrt(D / 3.0)
    x = None if c["same"] else rs.rand(N, D) * c["spread"] / np.sqrt(D / 3.0)
    if x is not None and c["offset"] > 0 and c["kernel"] != "exp-dot":
        u = rs.randn(D)
        x = x + c["offset"] * u / np.linalg.norm(u)
    if c["kernel"] == "exp-dot":  # logits of a few units to a few hundred
        g = float(rs.choice([1.0, 3.0, 10.0]))
        y = y * g
        x = None if x is None else x * g
    b = None if c["dens"] else rs.randn(M, E)
    if c["precision"] != "float64":
        # the inputs as the working precision sees them (bruteforce.py:97-99 casts them first): the sweep checks the
        # arithmetic, not what rounding a coordinate does to 1/r of nearly coincident points
        wp = np.float16 if c["precision"] == "float16" else np.float32
        y = y.astype(wp).astype(np.float64)
        x = None if x is None else x.astype(wp).astype(np.float64)
        b = None if b is None else b.astype(wp).astype(np.float64)
    algo = MI355XProduct(kernel=c["kernel"], dimension=D, normalize_rows=c["norm"], precision=c["precision"],
                         fast_sqdists=c["form"], fast_tiles=c["tiles"])
    try:
        algo.prepare_data(source_points=y, target_points=y if x is None else x, same_points=c["same"],
                          density_estimation=c["dens"])
        algo.fit()
        algo.prepare_query(source_signal=b)
        algo.query()
        got = algo.get_result()
        kname = algo.device_kernel
    finally:
        algo.done()
    rows = None
    if N * M > 20_000_000:  # the numpy oracle on a row sample
        rows = np.sort(rs.choice(N, size=256, replace=False))
        got = got[rows]
    if c["kernel"] == "exp-dot":
        return kname, check_exp_dot(c, got, y, x, b, rows)
    want = kmvp_oracle.product(kernel=c["kernel"], source_points=y, target_points=x, source_signal=b,
                               normalize_rows=c["norm"], density_estimation=c["dens"], rows=rows)
    finite = np.isfinite(want).all(axis=-1) if want.ndim > 1 else np.isfinite(want)
    if got.shape != want.shape:
        return kname, f"shape {got.shape} != {want.shape}"
    gf = np.isfinite(got).all(axis=-1) if got.ndim > 1 else np.isfinite(got)
    # fast_sqdists=True FORCED on exp(-r) or 1/r is the reference's expanded form with its absolute error in s
    # (bruteforce.py:36-49: cancellation near s = 0; the reference's own float32 run takes sqrt of negative numbers there).
    # Same rule as tests/test_gpu_parity.py::test_fast_sqdists_matches_reference: the yardstick is the REFERENCE's own
    # arithmetic in that form -- the oracle with precision=float32, fast_sqdists=True on the same inputs -- and the plugin
    # may be at most twice as far from the float64 truth; rows are compared where that oracle run is finite (where it is
    # not, the reference itself gives no answer to hold the plugin to).
    forced_expanded = c["form"] is True and c["kernel"] != "gaussian" and kname == "fast_kernel"
    ref_fast = None
    if forced_expanded:
        ref_fast = kmvp_oracle.product(kernel=c["kernel"], source_points=y, target_points=x, source_signal=b,
                                       normalize_rows=c["norm"], density_estimation=c["dens"], rows=rows,
                                       precision=np.float32, fast_sqdists=True)
        finite = finite & (np.isfinite(ref_fast).all(axis=-1) if ref_fast.ndim > 1 else np.isfinite(ref_fast))
        # ... and where the row's closest pair is resolved at all by an expansion around one centre: its s carries
        # ~1e-7 (|x - c|^2 + |y - c|^2) of absolute error, so a pair closer than that may come out as s <= 0 in one
        # float32 arithmetic and not in another (the reference's own run takes sqrt of negative numbers there)
        xt = y if x is None else x
        xt = xt if rows is None else xt[rows]
        cen = np.concatenate([y, xt]).mean(axis=0)
        r2 = ((xt - cen) ** 2).sum(axis=1).max() + ((y - cen) ** 2).sum(axis=1).max()
        smin = np.full(len(xt), np.inf)
        for j0 in range(0, len(y), 4096):
            yc, xc = y[j0:j0 + 4096] - cen, xt - cen  # (float64: the expanded form is exact enough for this bound)
            d = np.maximum((xc * xc).sum(axis=1)[:, None] + (yc * yc).sum(axis=1)[None, :] - 2.0 * (xc @ yc.T), 0.0)
            if x is None:
                d[d <= 1e-14 * r2] = np.inf  # the pair the index rule drops
            smin = np.minimum(smin, d.min(axis=1))
        finite = finite & (smin > 3e-4 * r2)  # 0.5 x 1e-7 r2 / s <= the 2e-4 the sweep holds 1/r to
    if c["offset"] > 0 and c["norm"] and c["precision"] != "float64":
        # a float32 denominator below the float32 range is 0 and the row 0/0, in the reference's float32 run as here
        # (the kernels with a per-target shift do better; the difference form does not have to)
        den = kmvp_oracle.product(kernel=c["kernel"], source_points=y, target_points=x, density_estimation=True, rows=rows)
        finite = finite & (den.reshape(len(den), -1) > 1e-30).all(axis=-1)
    if not np.array_equal(gf & finite, finite):
        return kname, f"non-finite rows differ: {int((gf & finite).sum())} finite of {int(finite.sum())} expected"
    if not finite.any():
        return kname, None
    scale = np.abs(want[finite]).max()
    if c["offset"] > 0 and not c["norm"] and scale > 0:
        # displaced targets: row by row against the row's own mass (a global yardstick would hide a row that lost its
        # digits); the float32 kernels flush rows below the float32 range to zero, those rows are left out
        mass = kmvp_oracle.product(kernel=c["kernel"], source_points=y, target_points=x,
                                   source_signal=None if b is None else np.abs(b), density_estimation=c["dens"], rows=rows)
        live = finite & ((mass > 1e-30).all(axis=-1) if mass.ndim > 1 else mass > 1e-30)
        if c["precision"] == "float64":
            live = finite
        if not live.any():
            return kname, None
        rel = float((np.abs(got[live] - want[live]) / np.maximum(mass[live], 1e-300)).max())
        tol = 1e-11 if c["precision"] == "float64" else (2e-4 if c["kernel"] == "inverse-distance" else 2e-5)
        if c["precision"] != "float64" and c["kernel"] != "inverse-distance":
            # float32 rounds s itself by 6e-8 s, which is an ABSOLUTE error of the exponent: s up to ~150 here
            xt = x if rows is None else x[rows]
            smax = float(((np.maximum(xt.max(axis=0), y.max(axis=0)) - np.minimum(xt.min(axis=0), y.min(axis=0))) ** 2).sum())
            tol += 3e-7 * (smax if c["kernel"] == "gaussian" else np.sqrt(smax))
        if forced_expanded:
            tol = max(tol, 2.0 * float((np.abs(ref_fast[live].astype(np.float64) - want[live]) / np.maximum(mass[live], 1e-300)).max()))
        if rel > tol and c["precision"] != "float64":
            ref32 = kmvp_oracle.product(kernel=c["kernel"], source_points=y, target_points=x, source_signal=b,
                                        density_estimation=c["dens"], rows=rows, precision=np.float32)
            tol = max(tol, 2.0 * float((np.abs(ref32[live] - want[live]) / np.maximum(mass[live], 1e-300)).max()))
        return kname, (None if rel <= tol else f"row-relative error {rel:.3e} > {tol:.0e}")
    err = float(np.abs(got[finite] - want[finite]).max() / scale) if scale > 0 else float(np.abs(got[finite]).max())
    tol = 1e-11 if c["precision"] == "float64" else 2e-5
    if c["kernel"] == "inverse-distance" and c["precision"] != "float64":
        tol = 2e-4  # 1/r of nearly coincident points
    if c["offset"] > 0 and c["precision"] != "float64" and c["kernel"] != "inverse-distance":
        # (normalised rows of displaced targets: float32 rounds s itself, an absolute error of the exponent, as above)
        xt = x if rows is None else x[rows]
        smax = float(((np.maximum(xt.max(axis=0), y.max(axis=0)) - np.minimum(xt.min(axis=0), y.min(axis=0))) ** 2).sum())
        tol += 4e-7 * (smax if c["kernel"] == "gaussian" else np.sqrt(smax))
    if forced_expanded and scale > 0:
        # (eight times here, twice in tests/ on the golden cases: both arithmetics round the same expansion, in different orders,
        # and on a handful of points the ratio of two such errors scatters -- seed 555 case 1704: 2.3 x, seed 9001 case 4453: 7 x
        # the reference's own float32 error, 5e-5 of the largest row; a wrong centre or a lost split would be 100 x)
        tol = max(tol, 8.0 * float(np.abs(ref_fast[finite].astype(np.float64) - want[finite]).max() / scale))
    if err > tol and c["precision"] != "float64" and scale > 0:
        # sums that cancel (|a| << sum |k b|) amplify every float32 rounding: the yardstick is then the reference's own
        # float32 arithmetic on the same inputs, as in tests/test_gpu_parity.py (max(tolerance, 2 x its error))
        ref32 = kmvp_oracle.product(kernel=c["kernel"], source_points=y, target_points=x, source_signal=b,
                                    normalize_rows=c["norm"], density_estimation=c["dens"], rows=rows,
                                    precision=np.float32)
        ok32 = finite & (np.isfinite(ref32).all(axis=-1) if ref32.ndim > 1 else np.isfinite(ref32))
        tol = max(tol, 2.0 * float(np.abs(ref32[ok32] - want[ok32]).max() / scale))
    return kname, (None if err <= tol else f"error {err:.3e} > {tol:.0e}")


def check_exp_dot(c, got, y, x, b, rows):
    """exp<x, y> against direct float64 evaluation (oracle/kmvp_oracle.py exp_dot_product), rows by their own mass."""
    xt = y if x is None else x
    if rows is not None:
        xt = xt[rows]
    if c["precision"] == "bfloat16":  # the truth on the operands the kernel multiplies: points x sqrt(log2 e), rounded to bf16
        k = 1.2011224087864498

        def bf16(a, c):  # ONE float32 product, as the packing kernel forms it, then round-to-nearest-even to bf16
            u = (np.ascontiguousarray(a, dtype=np.float32) * np.float32(c)).view(np.uint32)
            return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32).astype(np.float64) / c

        y, xt = bf16(y, k), bf16(xt, k)
    with np.errstate(over="ignore", invalid="ignore"):
        want = kmvp_oracle.exp_dot_product(source_points=y, target_points=xt, source_signal=b, normalize_rows=c["norm"])
        # the yardstick of a row is its mass: sum_j k |b_j|, or the weighted mean of |b| for a softmax row (means of both signs cancel)
        mass = kmvp_oracle.exp_dot_product(source_points=y, target_points=xt, source_signal=np.abs(b), normalize_rows=c["norm"])
    if got.shape != want.shape:
        return f"shape {got.shape} != {want.shape}"
    live = np.isfinite(want).all(axis=-1) & np.isfinite(mass).all(axis=-1)
    if c["precision"] != "float64":  # results beyond the float32 range are not representable in the working precision
        live &= (np.abs(mass) < 1e37).all(axis=-1)
    if not live.any():
        return None
    if not np.isfinite(got[live]).all():
        return f"non-finite rows: {int((~np.isfinite(got[live]).all(axis=-1)).sum())} of {int(live.sum())}"
    yard = np.maximum(np.abs(mass[live]), 1e-300)
    rel = float((np.abs(got[live] - want[live]) / yard).max())
    # float32: a logit carries ~1e-7 of the largest |x| |y| as ABSOLUTE error, which is a relative error of the weight
    lmax = float(np.sqrt((xt * xt).sum(axis=1).max() * (y * y).sum(axis=1).max()))
    tol = 1e-10 * max(1.0, lmax) if c["precision"] == "float64" else 2e-5 + 5e-7 * lmax
    if c["precision"] == "bfloat16":
        tol = 1e-2 + 5e-7 * lmax  # bf16 kernel values in the second product
    return None if rel <= tol else f"exp-dot error {rel:.3e} > {tol:.0e}"


def sweep(cases, seed, verbose=True):
    rs = np.random.RandomState(seed)
    seen, bad, failures = {}, 0, []
    for i in range(cases):
        c = one_case(rs)
        try:
            kname, msg = run_case(c, rs)
        except NotImplementedError as exc:  # e.g. float16 with an unsupported combination: as the reference's ctor
            kname, msg = "NotImplementedError", None
        except Exception as exc:  # noqa: BLE001
            kname, msg = "exception", f"{type(exc).__name__}: {exc}"
        seen[kname] = seen.get(kname, 0) + 1
        if msg:
            bad += 1
            failures.append(f"case {i} {c} -> {kname}: {msg}")
            if verbose:
                print("FAIL " + failures[-1], flush=True)
        if verbose and (i + 1) % 50 == 0:
            print(f"... {i + 1} cases, {bad} failures", flush=True)
    return seen, failures


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seen, failures = sweep(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n_cases} cases, {len(failures)} failures; kernels: {seen}")
    sys.exit(1 if failures else 0)
