"""Measures the BASELINE configs other than the headline one on the GPU box
(config 3: bf16 attention; config 4: one 1/8 source shard of inverse-distance at 1e7;
config 5: fp64 CG at 1e5).  Prints one line per config; numbers go into DESIGN.md."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kernel_matrix_benchmarks_amd import _lib
from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct, MI355XSolver
import c_oracle

which = sys.argv[1:] or ["3", "4", "5"]

if "3" in which:
    n, D, E = 65536, 64, 64
    rs = np.random.RandomState(n + D)
    y = rs.rand(n, D) / np.sqrt(D); b = rs.randn(n, E)
    rows = np.random.RandomState(0).choice(n, 256, replace=False)
    want = c_oracle.product(kernel="absolute-exponential", source_points=y, source_signal=b, normalize_rows=True, rows=rows)
    algo = MI355XProduct(kernel="absolute-exponential", dimension=D, normalize_rows=True, precision="bfloat16", segments=8)
    algo.prepare_data(source_points=y, target_points=y, same_points=True); algo.prepare_query(source_signal=b)
    algo.query(); ms = []
    for _ in range(10):
        t0 = time.perf_counter(); algo.query(); ms.append((time.perf_counter() - t0) * 1e3)
    a = algo.get_result()
    err = np.max(np.linalg.norm(a[rows] - want, axis=1)) / np.max(np.linalg.norm(want, axis=1))
    k = algo._ctx.last_kernel_ms
    flops = 2.0 * n * n * (80 + 64)
    print(f"config3 absexp attention bf16 N=M={n} D={D} E={E}: wall {min(ms):.3f} ms kernel {k:.3f} ms "
          f"pairs/s {n*n/(k*1e-3):.3e} MFMA {flops/(k*1e-3)/1e15:.3f} PFLOP/s ({flops/(k*1e-3)/2.5e15:.2%} of 2.5 PF) rel_err {err:.2e}", flush=True)
    algo.done()

if "4" in which:
    n = 10_000_000
    rs = np.random.RandomState(n + 3)
    y = rs.rand(n, 3); b = rs.randn(n, 1)
    lo, hi = 0, n // 8  # the shard of rank 0 of 8
    rows = np.random.RandomState(1).choice(n, 64, replace=False)
    want, _ = c_oracle.product(kernel="inverse-distance", source_points=y[lo:hi], target_points=y, source_signal=b[lo:hi], rows=rows, j_offset=lo, M_total=n, raw_sums=True)
    for form, opt in (("difference form", 0), ("auto", -1)):
        ctx = _lib.Context(0)
        ctx.set_option("fast_sqdists", opt)
        ctx.set_option("same_points_global", 1)  # the targets ARE the full source set (what the sharded plugin sets)
ctx.set_option("partial_shard", 1)  # a shard without a communicator, on purpose
        ctx.set_points(np.ascontiguousarray(y[lo:hi], dtype=np.float32), y.astype(np.float32), _lib.KMVP_F32, j_offset=lo, M_total=n)
        ctx.set_signal(np.ascontiguousarray(b[lo:hi], dtype=np.float32))
        ctx.run("inverse-distance", False)
        t0 = time.perf_counter(); ctx.run("inverse-distance", False); wall = time.perf_counter() - t0
        part = ctx.get_result(n, 1)
        err = np.max(np.abs(part[rows] - want)) / np.max(np.abs(want))
        print(f"config4 shard 1/8 ({form} -> {ctx.last_kernel_name}): N={n} x M={hi-lo} inverse-distance f32: wall {wall*1e3:.1f} ms kernel {ctx.last_kernel_ms:.1f} ms "
              f"pairs/s {n*(hi-lo)/(ctx.last_kernel_ms*1e-3):.3e} finite {bool(np.isfinite(part).all())} rel_err {err:.2e} device_MB {ctx.device_bytes/1e6:.0f}", flush=True)
        ctx.close()

if "5" in which:
    n = 100_000
    rs = np.random.RandomState(n + 3)
    y = rs.rand(n, 3); b = rs.randn(n, 1)
    prod = MI355XProduct(kernel="gaussian", dimension=3, precision=np.float64)
    prod.prepare_data(source_points=y, target_points=y, same_points=True); prod.prepare_query(source_signal=b)
    prod.query(); t0 = time.perf_counter(); prod.query(); tq = time.perf_counter() - t0
    a = prod.get_result()
    print(f"config5 operator: fp64 gaussian N=M={n}: {tq*1e3:.2f} ms per product, {n*n/tq:.3e} pairs/s (kernel {prod._ctx.last_kernel_ms:.2f} ms)", flush=True)
    prod.done()
    sol = MI355XSolver(kernel="gaussian", dimension=3, precision=np.float64, rtol=1e-6, maxit=5000)
    sol.prepare_data(source_points=y); sol.prepare_query(target_signal=a)
    t0 = time.perf_counter(); sol.query(); ts = time.perf_counter() - t0
    info = sol.get_additional()
    x = sol.get_result()
    rows = np.random.RandomState(2).choice(n, 256, replace=False)
    Kx = c_oracle.product(kernel="gaussian", source_points=y, source_signal=x, rows=rows)
    res = np.linalg.norm(Kx - a[rows]) / np.linalg.norm(a[rows])
    print(f"config5 CG fp64 N=M={n}: {ts:.2f} s, {info['cg_iterations']} iterations, residual {info['cg_relative_residual']:.2e} "
          f"(oracle on 256 rows: {res:.2e}), {info['cg_iterations']*n*n/ts:.3e} pairs/s", flush=True)
    sol.done()
