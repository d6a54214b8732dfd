"""Case list and seeded input generation shared by ``make_golden.py`` (which runs
the reference to produce ``golden/expected.npz``) and by the parity tests.

Inputs are regenerated from a seed with the legacy ``numpy.random.RandomState``
stream (frozen across numpy versions) following the reference's ``uniform_cube``
recipe, datasets.py:258-266: ``seed(n+D)``, ``rand(n,D)``, ``randn(n,E)``.  Only
the expected OUTPUTS are stored in the fixture file.
"""
import itertools
import numpy as np

KERNELS = ("gaussian", "absolute-exponential", "inverse-distance")


def make_inputs(case):
    """Returns (y (M,D), x (N,D) or None, b (M,E) or None) as float64."""
    N, M, D, E = case["N"], case["M"], case["D"], case["E"]
    rs = np.random.RandomState(case["seed"])
    scale = case.get("scale", 1.0)
    y = scale * rs.rand(M, D)
    b = rs.randn(M, E)
    if case["same_points"]:
        x = None
    else:
        x = scale * rs.rand(N, D)
    if case.get("duplicate"):
        # an off-diagonal coincident pair: inverse-distance gives inf there
        i, j = case["duplicate"]
        if x is None:
            y[j] = y[i]
        else:
            x[i] = y[j]
    if case["density_estimation"]:
        b = None
    return y, x, b


def _case(kernel, N, M, D, E, nr=False, sp=False, de=False, **kw):
    assert not sp or N == M
    name = f"{kernel}-N{N}-M{M}-D{D}-E{E}" + ("-nr" if nr else "") + ("-sp" if sp else "") + (
        "-de" if de else ""
    )
    for k, v in sorted(kw.items()):
        if k == "scale":
            name += f"-s{v:g}"
        elif k == "duplicate":
            name += "-dup"
    c = dict(
        name=name, kernel=kernel, N=N, M=M, D=D, E=(1 if de else E), normalize_rows=nr,
        same_points=sp, density_estimation=de, seed=M + D,
    )
    c.update(kw)
    return c


def product_cases():
    cases = []
    # 1. the reference's own shape family: D=3, E=1, same points (datasets.py:383-427)
    for kernel in KERNELS:
        for n in (1, 64, 1000):
            cases.append(_case(kernel, n, n, 3, 1, sp=True))
    # 2. ragged tails, x != y, both N<M and N>M (inverse-distance wrap pattern)
    for kernel in KERNELS:
        for (N, M) in ((257, 193), (193, 257), (64, 64), (1, 300), (300, 1)):
            cases.append(_case(kernel, N, M, 3, 1))
    # 3. flag combinations of query(): normalize_rows x density_estimation x same_points
    for kernel in KERNELS:
        for nr, sp, de in itertools.product((False, True), repeat=3):
            if (nr, sp, de) == (False, False, False):
                continue
            N, M = (193, 193) if sp else (257, 193)
            cases.append(_case(kernel, N, M, 3, 3, nr=nr, sp=sp, de=de))
    # 4. dimensions of the points and of the signal
    for kernel in KERNELS:
        for D in (1, 2, 4, 5, 8, 16):
            cases.append(_case(kernel, 130, 97, D, 1, scale=1.0 / np.sqrt(D)))
        for E in (2, 3, 4, 8):
            cases.append(_case(kernel, 130, 97, 3, E))
        cases.append(_case(kernel, 130, 97, 5, 3, nr=True))
        cases.append(_case(kernel, 64, 64, 16, 64, scale=0.25))
    # 5. high-D / wide-signal shapes (the attention-like tile), points scaled by 1/sqrt(D)
    for kernel in KERNELS:
        cases.append(_case(kernel, 96, 160, 64, 64, nr=True, scale=0.125))
        cases.append(_case(kernel, 96, 160, 64, 64, scale=0.125))
        cases.append(_case(kernel, 64, 64, 128, 16, nr=True, scale=1.0 / np.sqrt(128)))
    # 6. a coincident off-diagonal pair (inf for inverse-distance)
    cases.append(_case("inverse-distance", 64, 64, 3, 1, sp=True, duplicate=(5, 40)))
    cases.append(_case("inverse-distance", 70, 64, 3, 1, duplicate=(7, 9)))
    cases.append(_case("gaussian", 64, 64, 3, 1, sp=True, duplicate=(5, 40)))
    names = [c["name"] for c in cases]
    assert len(set(names)) == len(names), "duplicate case names"
    return cases


def solver_cases():
    return [
        dict(name="solver-gaussian-n64", kernel="gaussian", n=64, D=3, seed=67),
        dict(name="solver-gaussian-n500", kernel="gaussian", n=500, D=3, seed=503),
        dict(name="solver-absolute-exponential-n300", kernel="absolute-exponential", n=300, D=3, seed=303),
    ]


def make_solver_inputs(case):
    """y (n,D), b_true (n,1): the harness' solver datasets carry a = K b_true
    (datasets.py:133-195 with task='solver')."""
    rs = np.random.RandomState(case["seed"])
    y = rs.rand(case["n"], case["D"])
    b = rs.randn(case["n"], 1)
    return y, b
