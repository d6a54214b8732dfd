"""One rank of tests/test_gpu_multirank.py (started with RANK / WORLD_SIZE / MASTER_* in the environment).

Every rank drives the PLUGIN (MI355XProduct / MI355XSolver) on GPU 0 with the sources sharded over the ranks, through
the real libkmvp.so: shard-local pair loops with j_offset / M_total, the canonical unpadded exchange layout, the
all-reduce, normalisation, the sharded solvers.  The one thing that differs from a multi-GPU run is the wire: RCCL
refuses two ranks on one device, so the exchange is staged through host memory and summed by gloo
(include/kmvp.h kmvp_comm_init_host) -- test infrastructure, selected explicitly.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402


def main():
    from kernel_matrix_benchmarks_amd import _lib, sharding

    _lib.load()  # the system ROCm stack first (bench.py does the same)
    import torch.distributed as dist

    import kmvp_oracle
    from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct, MI355XSolver

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = sharding.torch_gloo_communicator(exchange="host")
    report = []

    def rel(a, b):
        return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

    # (kernel, normalize, precision, n, D, E, x != y, tolerance)
    products = [
        ("gaussian", False, "float32", 40000, 3, 1, False, 1e-5),          # cell order sharding, cell form per rank
        ("gaussian", True, "float32", 5000, 3, 5, True, 1e-5),             # several columns + denominator column
        ("inverse-distance", False, "float32", 20000, 3, 1, False, 2e-5),  # global zero rule, centred form (same_points_global)
        ("inverse-distance", True, np.float64, 3001, 2, 2, False, 1e-11),  # odd split, difference form fp64
        ("absolute-exponential", True, np.float64, 3000, 5, 3, True, 1e-11),
        ("absolute-exponential", True, "bfloat16", 4096, 64, 64, False, 1e-2),
        ("gaussian", False, "float32", world - 1, 3, 1, False, 1e-5),     # fewer sources than ranks: an EMPTY shard
    ]
    for kernel, normalize, precision, n, D, E, other, tol in products:
        rs = np.random.RandomState(n + D)
        y = rs.rand(n, D) / (np.sqrt(D) if D > 8 else 1.0)
        b = rs.randn(n, E)
        x = rs.rand(n // 2 + 3, D) / (np.sqrt(D) if D > 8 else 1.0) if other else None
        algo = MI355XProduct(kernel=kernel, dimension=D, normalize_rows=normalize, precision=precision, device=0, comm=comm)
        try:
            algo.prepare_data(source_points=y, target_points=y if x is None else x, same_points=x is None)
            algo.fit()
            algo.prepare_query(source_signal=b)
            algo.query()
            got = algo.get_result()
            meta = algo.get_additional()
            lo, hi = algo.shard
        finally:
            algo.done()
        want = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b, normalize_rows=normalize)
        e = rel(got, want)
        assert meta["rccl_ranks"] == world and meta["n_gpus"] == world, meta
        assert hi - lo == sharding.shard_range(n, rank, world)[1] - sharding.shard_range(n, rank, world)[0]
        assert np.isfinite(got).all() and e <= tol, (kernel, precision, n, e, tol, meta)
        report.append({"kernel": kernel, "precision": str(precision), "n": n, "shard": [lo, hi], "rel_err": e,
                       "device_kernel": meta["device_kernel"]})

    # exp(<x,y>) sharded: (mantissa, exponent) partial sums, all-reduce(min) of the exponents, then the sum
    rs = np.random.RandomState(5)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    y, x, b = f32(rs.randn(3001, 8) * 3.5), f32(rs.randn(500, 8)), f32(rs.randn(3001, 3))
    # (sources ordered by norm: the ranks' exponents then differ by hundreds)
    y = y[np.argsort(np.sum(y * y, axis=1))]
    for normalize, precision in ((True, "float32"), (False, "float32"), (True, "bfloat16"), (False, "bfloat16")):
        algo = MI355XProduct(kernel="exp-dot", dimension=8, normalize_rows=normalize, precision=precision, device=0, comm=comm)
        try:
            algo.prepare_data(source_points=y, target_points=x, same_points=False)
            algo.fit()
            algo.prepare_query(source_signal=b)
            algo.query()
            got = algo.get_result()
            meta = algo.get_additional()
        finally:
            algo.done()
        ys, xs = y, x
        if precision == "bfloat16":  # the truth on the operands the kernel multiplies (points x sqrt(log2 e), rounded to bf16)
            c = 1.2011224087864498

            def bf16(a):  # ONE float32 product, as the packing kernel forms it, rounded to bf16, divided by c again
                u = (np.ascontiguousarray(a, dtype=np.float32) * np.float32(c)).view(np.uint32)
                return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32).astype(np.float64) / c

            ys, xs = bf16(y), bf16(x)
        want = kmvp_oracle.exp_dot_product(source_points=ys, target_points=xs, source_signal=b, normalize_rows=normalize)
        mass = want if normalize else kmvp_oracle.exp_dot_product(source_points=ys, target_points=xs, source_signal=np.abs(b))
        scale = np.max(np.abs(mass), axis=1, keepdims=True)
        e = float(np.max(np.abs(got - want) / scale))
        native = "fastmm_kernel" if precision == "float32" else "mfma_pipe_kernel"
        assert meta["device_kernel"] == native and "online shift" in meta["dispatch_note"] and meta["rccl_ranks"] == world, meta
        assert np.isfinite(got).all() and e <= (1e-4 if precision == "float32" else 1e-2), ("exp-dot", precision, normalize, e)
        report.append({"kernel": "exp-dot", "precision": precision, "normalize": normalize, "rel_err": e,
                       "device_kernel": meta["device_kernel"]})

    # the bf16 Gaussian with targets far from every source, sharded: the ranks' shifts differ (sources sorted by distance), the
    # exponents are merged by all-reduce(min) exactly as for exp(<x,y>)
    rs = np.random.RandomState(6)
    c = 1.2011224087864498

    def bf16r(a):
        u = (np.ascontiguousarray(a, dtype=np.float32) * np.float32(c)).view(np.uint32)  # ONE float32 product, as the kernel forms it
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32).astype(np.float64) / c

    y = rs.rand(3001, 32) / np.sqrt(32 / 3.0)
    x = rs.rand(400, 32) / np.sqrt(32 / 3.0) + 14.0 / np.sqrt(32)
    y = y[np.argsort(-((y - x.mean(axis=0)) ** 2).sum(axis=1))]
    b = rs.randn(3001, 5)
    for normalize in (True, False):
        algo = MI355XProduct(kernel="gaussian", dimension=32, normalize_rows=normalize, precision="bfloat16", device=0, comm=comm)
        try:
            algo.prepare_data(source_points=y, target_points=x, same_points=False)
            algo.fit()
            algo.prepare_query(source_signal=b)
            algo.query()
            got = algo.get_result()
            meta = algo.get_additional()
        finally:
            algo.done()
        want = kmvp_oracle.product(kernel="gaussian", source_points=bf16r(y), target_points=bf16r(x), source_signal=b, normalize_rows=normalize)
        mass = kmvp_oracle.product(kernel="gaussian", source_points=bf16r(y), target_points=bf16r(x), source_signal=np.abs(b),
                                   normalize_rows=normalize)
        e = float(np.max(np.abs(got - want) / mass))
        assert "online shift" in meta["dispatch_note"] and meta["rccl_ranks"] == world, meta
        assert np.isfinite(got).all() and e <= 1e-2, ("bf16 gaussian far targets", normalize, e)
        report.append({"kernel": "gaussian", "precision": "bfloat16", "normalize": normalize, "rel_err": e,
                       "device_kernel": meta["device_kernel"]})

    # sharded solvers: replicated Krylov vectors, operator summed over the ranks in every iteration
    for kernel, n, rtol in (("gaussian", 3000, 1e-6), ("inverse-distance", 1500, 1e-8)):
        y, b = kmvp_oracle.uniform_cube(n, 3)
        if kernel == "inverse-distance":  # the reference's own solver datasets: well-conditioned (datasets.py:393-399)
            y = kmvp_oracle.uniform_sphere_points(n)
        a = kmvp_oracle.product(kernel=kernel, source_points=y, source_signal=b)
        algo = MI355XSolver(kernel=kernel, dimension=3, precision=np.float64, device=0, rtol=rtol, maxit=20000, comm=comm)
        try:
            algo.prepare_data(source_points=y)
            algo.fit()
            algo.prepare_query(target_signal=a)
            algo.query()
            sol = algo.get_result()
            meta = algo.get_additional()
        finally:
            algo.done()
        res = float(np.linalg.norm(kmvp_oracle.product(kernel=kernel, source_points=y, source_signal=sol) - a) / np.linalg.norm(a))
        assert meta["cg_converged"] and res <= 2 * rtol, (kernel, res, meta)
        report.append({"solver": kernel, "n": n, "iterations": meta["cg_iterations"], "true_residual": res})

    # every rank must hold the same answers: compare a digest
    import torch

    digest = torch.tensor([sum(r.get("rel_err", r.get("true_residual", 0.0)) for r in report)], dtype=torch.float64)
    lo_t, hi_t = digest.clone(), digest.clone()
    dist.all_reduce(lo_t, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi_t, op=dist.ReduceOp.MAX)
    assert float(lo_t[0]) == float(hi_t[0]), "ranks disagree on the results"
    if rank == 0:
        print(json.dumps({"world": world, "cases": report}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
