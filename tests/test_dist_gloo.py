"""CPU tests of the multi-rank path: two REAL processes (world_size 2, gloo) run the PLUGIN --
``MI355XProduct`` / ``MI355XSolver`` with ``comm=sharding.torch_gloo_communicator()`` -- end to end.

There is no GPU here, so the one thing replaced is the device context: ``_lib.Context`` becomes a
recorder that keeps what the plugin hands to the library (points, signal, options, communicator
bootstrap) and plays the library's part of the data flow with the oracle (test infrastructure):
partial sums of this rank's source slice -> one all-reduce of the (N, E[+1]) sums (gloo standing in
for RCCL) -> normalise.  Everything above the C ABI is the product's own code: shard ranges, the
cell-order permutation of the Gaussian's sources and of the signal, ``same_points_global``, the
unique-id hand-off, attaching a communicator to every NEW context (the runner builds and frees
several instances per definition, runner.py:70-93).
"""
import os
import socket
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(rank, world, port):
    for p in (HERE, os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..")):
        sys.path.insert(0, p)
    import torch.distributed as dist

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    return dist


def _install_recorder(dist, log):
    """Replaces the device context of this process by a recorder (see the module docstring)."""
    import torch

    import kmvp_oracle
    from kernel_matrix_benchmarks_amd import _lib

    class RecorderContext:
        live = 0

        def __init__(self, device=0):
            self.device = device
            self.comm_world = 0
            self.options = {}
            self.uid = None
            RecorderContext.live += 1
            log.append(("create", id(self)))

        # -- what the plugin calls ---------------------------------------------------------
        def set_option(self, key, value):
            self.options[key] = value

        def comm_init(self, unique_id, rank, world):
            assert len(unique_id) == _lib.UNIQUE_ID_BYTES
            self.uid, self.rank, self.world = bytes(unique_id), rank, world
            self.comm_world = world
            log.append(("comm_init", id(self)))

        def set_points(self, y, x, dtype_code_, j_offset=0, M_total=None):
            assert self.comm_world == dist.get_world_size(), "sharded points on a context without a communicator"
            self.y, self.x = y.copy(), (None if x is None else x.copy())
            self.j_offset, self.M_total, self.dtype = j_offset, (len(y) if M_total is None else M_total), dtype_code_
            self.M, self.N = len(y), (len(y) if x is None else len(x))

        def fit(self, kernel):
            log.append(("fit", kernel))

        def set_signal(self, b):
            assert b is None or b.shape[0] == self.M
            self.b = None if b is None else b.copy()

        def _apply(self, kernel, signal, normalize_rows=False):
            num, den = kmvp_oracle.product(kernel=kernel, source_points=self.y.astype(np.float64),
                                           target_points=self.x.astype(np.float64), source_signal=signal,
                                           j_offset=self.j_offset, M_total=self.M_total, raw_sums=True)
            sums = torch.from_numpy(np.ascontiguousarray(np.concatenate([num, den], axis=1)))
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)  # libkmvp: ONE ncclAllReduce of the (N, E+1) sums
            sums = sums.numpy()
            return sums[:, :-1] / sums[:, -1:] if normalize_rows else sums[:, :-1]

        def run(self, kernel, normalize_rows):
            self.result = self._apply(kernel, self.b.astype(np.float64), normalize_rows)
            self.last_kernel_name = "recorder"

        def get_result(self, N, E):
            assert self.result.shape == (N, E)
            return np.ascontiguousarray(self.result, dtype=np.float64)

        def cg_solve(self, kernel, a, rtol, maxit):
            # kmvp_solvers.hip: Krylov vectors replicated on every rank, this rank's signal is its own slice
            # [j_offset, j_offset + M) of the replicated vector, sums all-reduced by the product
            assert self.options.get("same_points_global") == 1 and self.N == self.M_total
            lo, hi = self.j_offset, self.j_offset + self.M
            a = a.astype(np.float64)
            x = np.zeros_like(a)
            r = a.copy()
            p = a.copy()
            rs_old = float((r * r).sum())
            a2 = rs_old
            it = 0
            while it < maxit and np.sqrt(rs_old / a2) > rtol:
                Ap = self._apply(kernel, p[lo:hi])
                alpha = rs_old / float((p * Ap).sum())
                x += alpha * p
                r -= alpha * Ap
                rs_new = float((r * r).sum())
                p = r + (rs_new / rs_old) * p
                rs_old = rs_new
                it += 1
            true = np.linalg.norm(a - self._apply(kernel, x[lo:hi])) / np.sqrt(a2)
            return x, it, float(true), bool(true <= 1.5 * rtol)

        def close(self):
            if not getattr(self, "closed", False):
                self.closed = True
                RecorderContext.live -= 1
                log.append(("close", id(self)))

        device_bytes = 0
        last_kernel_ms = last_total_ms = last_allreduce_ms = 0.0
        last_kernel_name = "recorder"
        last_dispatch_note = ""

        @property
        def rccl_ranks(self):
            return self.comm_world or 1

    _lib.Context = RecorderContext
    _lib.comm_unique_id = lambda: bytes(range(128))[::-1]  # rank 0's payload; RCCL itself needs a GPU
    return RecorderContext


def _product_worker(rank, world, port, out_dir):
    dist = _setup(rank, world, port)
    try:
        import golden_cases
        import kmvp_oracle
        from kernel_matrix_benchmarks_amd import sharding
        from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct

        log = []
        Recorder = _install_recorder(dist, log)
        comm = sharding.torch_gloo_communicator()
        assert (comm.rank, comm.world) == (rank, world)
        results = {}
        rs = np.random.RandomState(17)
        n = 1500
        y = rs.rand(n, 3) * 1.3 + 0.2
        b = rs.randn(n, 2)
        # same_points (what every reference dataset is, datasets.py:238,276): two instances IN A ROW per
        # kernel, the first one freed before the second exists -- the id()-reuse pattern of runner.run
        for kernel in golden_cases.KERNELS:
            for normalize in (False, True):
                for instance in range(2):
                    algo = MI355XProduct(kernel=kernel, dimension=3, normalize_rows=normalize, precision="float32",
                                         comm=comm)
                    algo.prepare_data(source_points=y, target_points=y, same_points=np.bool_(True))
                    algo.fit()
                    ctx = algo._ctx
                    assert isinstance(ctx, Recorder) and ctx.comm_world == world, "fresh context without a communicator"
                    assert ctx.uid == bytes(range(128))[::-1]  # rank 0's unique id reached this rank
                    lo, hi = sharding.shard_range(n, rank, world)
                    assert algo.shard == (lo, hi) and (ctx.j_offset, ctx.M_total) == (lo, n)
                    y32 = y.astype(np.float32)
                    order = sharding.spatial_order(y32) if kernel == "gaussian" else None
                    # index-aligned targets and sources (the inverse-distance zero rule, bruteforce.py:13-14) only
                    # where the sources kept the caller's order
                    assert ctx.options.get("same_points_global") == (1 if order is None else 0)
                    idx = np.arange(n) if order is None else order
                    assert np.array_equal(ctx.y, y32[idx][lo:hi])  # this rank's slice (cell order for the Gaussian)
                    assert np.array_equal(ctx.x, y32)             # all targets, caller's order
                    algo.prepare_query(source_signal=b)
                    assert np.array_equal(ctx.b, b.astype(np.float32)[idx][lo:hi])  # signal permuted the same way
                    algo.query()
                    got = algo.get_result()
                    extra = algo.get_additional()
                    assert extra["n_gpus"] == world and extra["rccl_ranks"] == world
                    algo.done()
                    del algo
                    results[f"{kernel}-{int(normalize)}-{instance}"] = got
                    # every rank derived the same permutation and the slices partition the sources
                    mine = np.full(n, -1, dtype=np.int64)
                    mine[: hi - lo] = idx[lo:hi]
                    import torch

                    gathered = [torch.empty(n, dtype=torch.int64) for _ in range(world)]
                    dist.all_gather(gathered, torch.from_numpy(mine))
                    owned = np.concatenate([g.numpy()[g.numpy() >= 0] for g in gathered])
                    assert sorted(owned.tolist()) == list(range(n)), "slices are not a partition of the sources"
        assert Recorder.live == 0
        creates = sum(1 for e in log if e[0] == "create")
        inits = sum(1 for e in log if e[0] == "comm_init")
        assert creates == inits == 12, (creates, inits)  # every NEW context got its communicator

        # targets != sources, ragged shapes, E = 3: the golden N < M and N > M cases
        for kernel in golden_cases.KERNELS:
            for (N, M) in ((257, 193), (193, 257)):
                case = dict(N=N, M=M, D=3, E=3, seed=M + 3, same_points=False, density_estimation=False)
                yy, xx, bb = golden_cases.make_inputs(case)
                algo = MI355XProduct(kernel=kernel, dimension=3, normalize_rows=False, precision=np.float64, comm=comm)
                algo.prepare_data(source_points=yy, target_points=xx, same_points=False)
                assert algo._ctx.options.get("same_points_global") == 0
                algo.prepare_query(source_signal=bb)
                algo.query()
                results[f"{kernel}-{N}-{M}"] = algo.get_result()
                algo.done()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **results)
    finally:
        dist.destroy_process_group()


def test_two_rank_plugin_product_equals_unsharded(tmp_path):
    import torch.multiprocessing as mp

    import golden_cases
    import kmvp_oracle

    mp.spawn(_product_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert sorted(r0.files) == sorted(r1.files) and len(r0.files) == 18
    for key in r0.files:
        assert np.array_equal(r0[key], r1[key]), key  # the all-reduce hands every rank the same sums
    rs = np.random.RandomState(17)
    y = rs.rand(1500, 3) * 1.3 + 0.2
    b = rs.randn(1500, 2)
    y32, b32 = y.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64)
    for kernel in golden_cases.KERNELS:
        for normalize in (False, True):
            want = kmvp_oracle.product(kernel=kernel, source_points=y32, source_signal=b32, normalize_rows=normalize)
            for instance in range(2):
                np.testing.assert_allclose(r0[f"{kernel}-{int(normalize)}-{instance}"], want, rtol=1e-11, atol=1e-11)
        for (N, M) in ((257, 193), (193, 257)):
            case = dict(N=N, M=M, D=3, E=3, seed=M + 3, same_points=False, density_estimation=False)
            yy, xx, bb = golden_cases.make_inputs(case)
            full = kmvp_oracle.product(kernel=kernel, source_points=yy, target_points=xx, source_signal=bb)
            np.testing.assert_allclose(r0[f"{kernel}-{N}-{M}"], full, rtol=1e-12, atol=1e-12)


def _solver_worker(rank, world, port, out_dir):
    dist = _setup(rank, world, port)
    try:
        from kernel_matrix_benchmarks_amd import sharding
        from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XSolver

        log = []
        Recorder = _install_recorder(dist, log)
        comm = sharding.torch_gloo_communicator()
        n, D = 301, 3
        rs = np.random.RandomState(n + D)
        y = rs.rand(n, D)
        a = rs.randn(n, 1)
        out = {}
        for instance in range(2):  # twice in a row: the second context must get its own communicator
            algo = MI355XSolver(kernel="absolute-exponential", dimension=D, precision=np.float64, rtol=1e-9,
                                maxit=2000, comm=comm)
            algo.prepare_data(source_points=y)
            algo.fit()
            ctx = algo._ctx
            lo, hi = sharding.shard_range(n, rank, world)
            assert isinstance(ctx, Recorder) and ctx.comm_world == world
            assert np.array_equal(ctx.y, y[lo:hi]) and np.array_equal(ctx.x, y)  # never reordered: Krylov slices
            assert (ctx.j_offset, ctx.M_total) == (lo, n)
            algo.prepare_query(target_signal=a)
            algo.query()
            out[f"x{instance}"] = algo.get_result()
            info = algo.get_additional()
            assert info["cg_converged"] and info["n_gpus"] == world
            out[f"it{instance}"] = info["cg_iterations"]
            algo.done()
        assert sum(1 for e in log if e[0] == "comm_init") == 2
        np.savez(os.path.join(out_dir, f"solver_rank{rank}.npz"), **out)
    finally:
        dist.destroy_process_group()


def test_two_rank_plugin_solver_iterates_in_lockstep(tmp_path):
    import torch.multiprocessing as mp

    import kmvp_oracle

    mp.spawn(_solver_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "solver_rank0.npz")
    r1 = np.load(tmp_path / "solver_rank1.npz")
    assert int(r0["it0"]) == int(r1["it0"]) == int(r0["it1"]) > 0
    assert np.array_equal(r0["x0"], r1["x0"]) and np.array_equal(r0["x0"], r0["x1"])
    n, D = 301, 3
    rs = np.random.RandomState(n + D)
    y = rs.rand(n, D)
    a = rs.randn(n, 1)
    Kx = kmvp_oracle.product(kernel="absolute-exponential", source_points=y, source_signal=r0["x0"])
    assert np.linalg.norm(Kx - a) / np.linalg.norm(a) < 1e-7  # exp(-r): well conditioned


def test_attach_follows_the_context_not_its_id():
    """Communicator.attach used to remember id(ctx); a freed context's id is reused by CPython."""
    from kernel_matrix_benchmarks_amd import _lib, sharding

    sent = []

    def bcast(payload):
        sent.append(payload)
        return payload if payload is not None else b"\0" * 128

    class Ctx:
        def __init__(self):
            self.comm_world = 0
            self.inits = 0

        def comm_init(self, uid, rank, world):
            self.inits += 1
            self.comm_world = world

    real = _lib.comm_unique_id
    _lib.comm_unique_id = lambda: b"\1" * 128
    try:
        comm = sharding.Communicator(0, 2, bcast)
        seen_ids = set()
        for _ in range(50):  # ids do repeat within a few allocations
            c = Ctx()
            seen_ids.add(id(c))
            comm.attach(c)
            comm.attach(c)  # idempotent for the SAME context
            assert c.inits == 1
            del c
        assert len(seen_ids) < 50, "this interpreter never reused an id: the test lost its point"
        assert len(sent) == 50
        one = sharding.Communicator(0, 1, bcast)
        c = Ctx()
        one.attach(c)
        assert c.inits == 0  # a single rank needs no communicator
    finally:
        _lib.comm_unique_id = real
