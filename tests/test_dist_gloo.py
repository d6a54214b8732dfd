"""CPU test of the multi-rank path: world_size 2 over gloo.

Each rank takes its source shard (sharding.shard_range), forms the partial sums its
GPU would form (here with the oracle, since there is no GPU), the partial (N, E+1)
sums are all-reduced, then normalised -- exactly the data flow of libkmvp.so with an
RCCL communicator attached (pair loop -> reduce_segments -> ncclAllReduce -> finish).
The unique-id hand-off of sharding.Communicator is exercised with a fake context.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    sys.path.insert(0, os.path.join(HERE, ".."))
    import torch
    import torch.distributed as dist

    import golden_cases
    import kmvp_oracle
    from kernel_matrix_benchmarks_amd import sharding

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        comm = sharding.torch_gloo_communicator()
        assert (comm.rank, comm.world) == (rank, world)

        # unique-id hand-off: rank 0's payload must arrive everywhere
        got = comm._broadcast(b"id-from-rank-0" if rank == 0 else None)
        assert got == b"id-from-rank-0"

        results = {}
        for kernel in golden_cases.KERNELS:
            for (N, M) in ((257, 193), (193, 257)):
                case = dict(N=N, M=M, D=3, E=3, seed=M + 3, same_points=False, density_estimation=False)
                y, x, b = golden_cases.make_inputs(case)
                lo, hi = sharding.shard_range(M, rank, world)
                num, den = kmvp_oracle.product(
                    kernel=kernel, source_points=y[lo:hi], target_points=x, source_signal=b[lo:hi],
                    j_offset=lo, M_total=M, raw_sums=True)
                sums = torch.from_numpy(np.concatenate([num, den], axis=1))
                dist.all_reduce(sums, op=dist.ReduceOp.SUM)
                sums = sums.numpy()
                results[f"{kernel}-{N}-{M}-prod"] = sums[:, :-1]
                results[f"{kernel}-{N}-{M}-norm"] = sums[:, :-1] / sums[:, -1:]
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **results)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_product_equals_unsharded(tmp_path):
    import socket

    import torch.multiprocessing as mp

    import golden_cases
    import kmvp_oracle

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for kernel in golden_cases.KERNELS:
        for (N, M) in ((257, 193), (193, 257)):
            case = dict(N=N, M=M, D=3, E=3, seed=M + 3, same_points=False, density_estimation=False)
            y, x, b = golden_cases.make_inputs(case)
            full = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x, source_signal=b)
            normed = kmvp_oracle.product(kernel=kernel, source_points=y, target_points=x,
                                         source_signal=b, normalize_rows=True)
            for r in (r0, r1):
                np.testing.assert_allclose(r[f"{kernel}-{N}-{M}-prod"], full, rtol=1e-12, atol=1e-12)
                np.testing.assert_allclose(r[f"{kernel}-{N}-{M}-norm"], normed, rtol=1e-12, atol=1e-12)
            assert np.array_equal(r0[f"{kernel}-{N}-{M}-prod"], r1[f"{kernel}-{N}-{M}-prod"])


def _solver_worker(rank, world, port, out_dir):
    """The sharded solver's data flow (kmvp_solvers.hip with a communicator): Krylov vectors
    replicated, operator = own source slice x all targets, partial sums all-reduced."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    sys.path.insert(0, os.path.join(HERE, ".."))
    import torch
    import torch.distributed as dist

    import kmvp_oracle
    from kernel_matrix_benchmarks_amd import sharding

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        n, D = 301, 3
        rs = np.random.RandomState(n + D)
        y = rs.rand(n, D)
        a = rs.randn(n, 1)
        lo, hi = sharding.shard_range(n, rank, world)

        def apply(v):  # cg_apply: this rank's slice of the replicated vector is the shard's signal
            part, _ = kmvp_oracle.product(kernel="absolute-exponential", source_points=y[lo:hi], target_points=y,
                                          source_signal=v[lo:hi], j_offset=lo, M_total=n, raw_sums=True)
            t = torch.from_numpy(np.ascontiguousarray(part))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.numpy()

        x = np.zeros_like(a); r = a.copy(); p = a.copy()
        rs_old = float((r * r).sum()); a2 = rs_old
        it = 0
        while it < 2000 and np.sqrt(rs_old / a2) > 1e-9:
            Ap = apply(p)
            alpha = rs_old / float((p * Ap).sum())
            x += alpha * p
            r -= alpha * Ap
            rs_new = float((r * r).sum())
            p = r + (rs_new / rs_old) * p
            rs_old = rs_new
            it += 1
        np.savez(os.path.join(out_dir, f"solver_rank{rank}.npz"), x=x, iterations=it)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_solver_iterates_in_lockstep(tmp_path):
    import socket

    import torch.multiprocessing as mp

    import kmvp_oracle

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_solver_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "solver_rank0.npz")
    r1 = np.load(tmp_path / "solver_rank1.npz")
    assert int(r0["iterations"]) == int(r1["iterations"]) > 0
    assert np.array_equal(r0["x"], r1["x"])  # the all-reduce hands every rank the same sums
    n, D = 301, 3
    rs = np.random.RandomState(n + D)
    y = rs.rand(n, D)
    a = rs.randn(n, 1)
    Kx = kmvp_oracle.product(kernel="absolute-exponential", source_points=y, source_signal=r0["x"])
    assert np.linalg.norm(Kx - a) / np.linalg.norm(a) < 1e-7  # exp(-r): well conditioned, recurrence and true residual agree
