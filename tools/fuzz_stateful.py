"""Stateful sweep at the C ABI: ONE context lives through a random sequence of kmvp_set_points / kmvp_set_signal / kmvp_fit /
option changes / products of DIFFERENT kernel functions, and every product is checked against the float64 numpy oracle.
What it is after: stale packed layouts, buffers sized for an earlier shape, shifts / exponents / cell lists left over from
another kernel function -- anything a caller of include/kmvp.h can reach by calling the entry points in an unusual order.
usage: python tools/fuzz_stateful.py [contexts=30] [steps=40] [seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd import _lib  # noqa: E402
import kmvp_oracle  # noqa: E402  (checker only)

KERNELS = ("gaussian", "absolute-exponential", "inverse-distance", "exp-dot")
C_DOT = 1.2011224087864498


def bf16(a, c):
    """(float32 a) x (float32 c) as ONE float32 product -- what the packing kernels form -- rounded to bf16, divided by c again"""
    u = (np.ascontiguousarray(a, dtype=np.float32) * np.float32(c)).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32).astype(np.float64) / c


def truth(kernel, y, x, b, norm, precision):
    if kernel == "exp-dot":
        ys, xs = y, (y if x is None else x)
        if precision == "bfloat16":
            ys, xs = bf16(ys, C_DOT), bf16(xs, C_DOT)
        with np.errstate(over="ignore", invalid="ignore"):
            want = kmvp_oracle.exp_dot_product(source_points=ys, target_points=xs, source_signal=b, normalize_rows=norm)
            mass = kmvp_oracle.exp_dot_product(source_points=ys, target_points=xs, source_signal=np.abs(b), normalize_rows=norm)
        return want, mass, None
    if precision == "bfloat16":  # the operands the kernel multiplies: points x the kernel's constant, rounded to bf16
        k = {"gaussian": C_DOT, "absolute-exponential": 1.4426950408889634, "inverse-distance": 1.0}[kernel]
        y = bf16(y, k)
        x = None if x is None else bf16(x, k)
    want = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b, normalize_rows=norm)
    # the yardstick of a row is its mass sum_j k |b_j| (normalised rows: the weighted mean of |b|): sums of both signs cancel
    mass = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=np.abs(b), normalize_rows=norm)
    den = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, density_estimation=True)
    return want, mass, den.reshape(len(den), -1)[:, 0]


def one_context(rs, steps, verbose):
    precision = ["float32", "float32", "float64", "bfloat16"][rs.randint(4)]
    code, host = _lib.dtype_code(precision)
    ctx = _lib.Context(0)
    failures, done = [], 0
    y = x = b = None
    state = {}
    try:
        for step in range(steps):
            action = rs.choice(["points", "signal", "option", "run", "run", "run"]) if y is not None and b is not None else \
                ("points" if y is None else "signal")
            if action == "points":
                D = int(rs.choice([1, 2, 3, 3, 4, 6, 20] if precision != "bfloat16" else [16, 24, 64, 100]))
                M = int(rs.choice([1, 31, 33, 200, 1000, 5000, 40000]))
                same = bool(rs.rand() < 0.4)
                N = M if same else int(rs.choice([1, 17, 64, 300, 3000, 40000]))
                spread = float(rs.choice([0.3, 1.0, 2.5]))
                y = (rs.rand(M, D) * spread / np.sqrt(D / 3.0)).astype(host).astype(np.float64)
                x = None if same else (rs.rand(N, D) * spread / np.sqrt(D / 3.0) + (rs.rand() < 0.2) * 3.0).astype(host).astype(np.float64)
                ctx.set_points(np.ascontiguousarray(y, dtype=host), None if x is None else np.ascontiguousarray(x, dtype=host), code)
                b = None  # a new source set needs a new signal
                state = dict(D=D, M=M, N=N, same=same)
            elif action == "signal":
                E = int(rs.choice([1, 1, 2, 4, 5, 16, 33]))
                b = rs.randn(state["M"], E).astype(host).astype(np.float64)
                ctx.set_signal(np.ascontiguousarray(b, dtype=host))
                state["E"] = E
            elif action == "option":
                key = str(rs.choice(["fast_sqdists", "fast_tiles", "segments", "cellmm_shape"]))
                val = {"fast_sqdists": [-1, -1, 0, 1, 2, 3, 4], "fast_tiles": [0, 0, 1, 2, 4, 8], "segments": [0, 0, 1, 3, 8],
                       "cellmm_shape": [-1, 0, 1]}[key]
                ctx.set_option(key, int(rs.choice(val)))
                if rs.rand() < 0.3:
                    ctx.fit("gaussian")
            else:
                kernel = KERNELS[rs.randint(4)]
                norm = bool(rs.rand() < 0.4)
                if kernel == "exp-dot" and (precision == "float64" or (precision == "float32" and state["D"] > 64)):
                    continue
                if state["N"] * state["M"] > 3e8:
                    continue
                try:
                    ctx.run(kernel, norm)
                except _lib.KmvpError as exc:
                    if "UNSUPPORTED" in str(exc).upper() or "unsupported" in str(exc):
                        continue
                    failures.append(f"step {step} {precision} {kernel} norm={norm} {state}: {exc}")
                    continue
                got = ctx.get_result(state["N"], state["E"])
                kname = ctx.last_kernel_name
                want, mass, den = truth(kernel, y, x, b, norm, precision)
                live = np.isfinite(want).all(axis=1) & np.isfinite(mass).all(axis=1)
                if precision != "float64":
                    live &= (np.abs(mass) < 1e37).all(axis=1) & ((np.abs(mass) > 1e-30).all(axis=1) | norm)
                if precision == "bfloat16" and den is not None and kernel != "gaussian":
                    # bf16 exp(-r) and 1/r carry no running shift (the Gaussian with targets != sources and exp<x, y> do): a row
                    # whose kernel values all lie under the float32 range is 0 (0/0 when normalised), as in the reference's float32
                    live &= den > 1e-30
                done += 1
                if not live.any():
                    continue
                if not np.isfinite(got[live]).all():
                    failures.append(f"step {step} {precision} {kernel} norm={norm} {state} -> {kname}: non-finite rows")
                    continue
                err = float((np.abs(got[live] - want[live]) / np.maximum(np.abs(mass[live]), 1e-300)).max())
                tol = {"float64": 1e-10, "float32": 2e-4 if kernel == "inverse-distance" else 5e-5, "bfloat16": 2e-2}[precision]
                if err > tol:
                    failures.append(f"step {step} {precision} {kernel} norm={norm} {state} -> {kname}: error {err:.3e} > {tol:.0e}")
    finally:
        ctx.close()
    if verbose:
        for f in failures:
            print("FAIL " + f, flush=True)
    return done, failures


if __name__ == "__main__":
    contexts = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rs = np.random.RandomState(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    total, bad = 0, []
    for c in range(contexts):
        done, failures = one_context(rs, steps, True)
        total += done
        bad += failures
        if (c + 1) % 10 == 0:
            print(f"... {c + 1} contexts, {total} products checked, {len(bad)} failures", flush=True)
    print(f"{contexts} contexts, {total} products checked, {len(bad)} failures")
    sys.exit(1 if bad else 0)
