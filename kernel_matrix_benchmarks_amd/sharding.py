"""Source sharding over the GPUs of one node.

One process per GPU.  Rank r owns the contiguous source slice
``shard_range(M, r, world)`` and all targets; the (N, E[+1]) partial sums are
summed with ONE RCCL all-reduce inside ``libkmvp.so`` (``kmvp_comm_init`` /
``ncclAllReduce`` on the context's stream).  This module only does the host-side
bookkeeping: the slice arithmetic and handing the 128-byte RCCL unique id from
rank 0 to the other ranks through an out-of-band channel.

The reference has no distributed code at all; nothing here mirrors a
reference file.
"""


def shard_range(M, rank, world):
    """Contiguous [lo, hi) of the M sources owned by ``rank``; sizes differ by <= 1."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(int(M), world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


CELL_T_MAX = 0.016  # kmvp_cell.hpp: bound on |2 d.e| that fixes the cell side


def cell_indices(p):
    """Integer cell coordinates of float32 points on libkmvp's grid (kmvp_product.hip:cell_make_grid): per axis
    the bounding box is divided into the smallest number of equal cells of side <= sqrt(2 CELL_T_MAX / D).
    ``None`` when an axis would need more than 1024 cells."""
    import numpy as np

    h_max = np.sqrt(2.0 * CELL_T_MAX / p.shape[1])
    lo = p.min(axis=0).astype(np.float64)
    extent = p.max(axis=0).astype(np.float64) - lo
    counts = np.maximum(1.0, np.ceil(extent / h_max))
    if counts.max() > 1024:
        return None
    h = np.where(extent > 0, extent / counts, h_max)
    cells = np.floor((p.astype(np.float64) - lo) / h).astype(np.int64)
    return np.minimum(cells, counts.astype(np.int64) - 1)


def spatial_order(points):
    """Permutation that lists the points cell by cell of a regular grid (the grid of libkmvp's
    ``cell_kernel``: side sqrt(2 * CELL_T_MAX / D)), or ``None`` when that grid does not apply (D > 3,
    non-finite coordinates, more than 1024 cells along an axis).

    A sum over sources does not care about their order, so a sharded Gaussian product may hand rank r
    the r-th slice of the sources *in this order* instead of in the caller's: each rank then holds
    whole cells (a few hundred sources per cell) rather than an eighth of every cell, which is what
    keeps the per-cell work of ``cell_kernel`` amortised.  Kernels with an index-based rule
    (inverse-distance: bruteforce.py:13-14) keep the caller's order.
    """
    import numpy as np

    p = np.asarray(points, dtype=np.float32)
    if p.ndim != 2 or p.shape[0] == 0 or p.shape[1] > 3 or not np.isfinite(p).all():
        return None
    cells = cell_indices(p)
    if cells is None:
        return None
    key = np.zeros(p.shape[0], dtype=np.int64)
    for a in range(p.shape[1]):
        key |= cells[:, a] << (10 * a)
    return np.argsort(key, kind="stable")


class Communicator:
    """Binds contexts of this process to an RCCL communicator.

    ``broadcast_bytes(payload_or_None) -> bytes`` must return rank 0's payload on
    every rank (e.g. ``torch.distributed.broadcast_object_list`` over gloo, or an
    MPI bcast); it is only used to distribute the RCCL unique id.
    """

    def __init__(self, rank, world, broadcast_bytes, host_allreduce=None):
        self.rank = int(rank)
        self.world = int(world)
        self._broadcast = broadcast_bytes
        # REHEARSAL only (several ranks on ONE GPU, where RCCL refuses to form a communicator): a callable
        # (array, op) that reduces a float64 numpy array in place over the ranks (op 0: sum, 1: min); the library
        # then stages its exchange through host memory
        # (include/kmvp.h kmvp_comm_init_host) instead of calling ncclAllReduce
        self._host_allreduce = host_allreduce

    def attach(self, ctx):
        """Collective: every rank calls it for its context.  Whether a context already has its communicator
        is recorded on the context object itself (``ctx.comm_world``), never by ``id(ctx)``: the runner builds
        and frees several instances per definition and CPython hands a freed object's id to the next one, which
        would skip the bootstrap for a fresh context and return its un-reduced partial sums as the result."""
        from kernel_matrix_benchmarks_amd import _lib

        if self.world == 1 or getattr(ctx, "comm_world", 0) == self.world:
            return
        if self._host_allreduce is not None:
            ctx.comm_init_host(self._host_allreduce, self.rank, self.world)
            return
        uid = _lib.comm_unique_id() if self.rank == 0 else None
        uid = self._broadcast(uid)
        ctx.comm_init(uid, self.rank, self.world)
        if getattr(ctx, "comm_world", None) != self.world:  # contexts that do not record it themselves
            ctx.comm_world = self.world


def torch_gloo_communicator(exchange="rccl"):
    """Communicator over an initialised ``torch.distributed`` process group (any
    backend that can broadcast Python objects from the host, i.e. gloo).  The group only carries the RCCL
    unique id; the sums travel over RCCL inside the library.  ``exchange="host"`` (rehearsal on a one-GPU
    box): the sums are staged through host memory and summed by this group instead."""
    import torch.distributed as dist

    def bcast(payload):
        box = [payload]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    host = None
    if exchange == "host":
        import torch

        def host(array, op):
            dist.all_reduce(torch.from_numpy(array), op=dist.ReduceOp.MIN if op == 1 else dist.ReduceOp.SUM)  # in place

    elif exchange != "rccl":
        raise ValueError("exchange must be 'rccl' or 'host'")
    return Communicator(dist.get_rank(), dist.get_world_size(), bcast, host_allreduce=host)
