#!/usr/bin/env python3
"""Headline benchmark: point-pair interactions per second of the on-the-fly kernel
matrix-vector product on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): Gaussian product, uniform points in the unit
cube (the reference's ``uniform_cube`` recipe, datasets.py:256-266, seed n+D),
N = M = 1e6, D = 3, E = 1, float32.  One "step" = one full product a = K b
(1e12 point pairs) with the points and the signal already resident in HBM: exactly
what the harness times around ``query()`` (runner.py:138-140); ``fit()`` only sorts the
points into grid cells (~3 ms, once), so query time ~ total time.

With N > 1 GPUs (launched by ``python -m torch.distributed.run``, one rank per GPU)
the M sources are sharded over the ranks and the (N, E) partial sums are summed by
one RCCL all-reduce inside every step: total work is fixed -> "scaling": "strong".

Rank 0 prints ONE JSON line.  ``roofline`` describes the dominant kernel
(lowd_kernel) from HIP-event timings of every timed launch; ``cpu_baseline`` is the
C/OpenMP restatement of the reference arithmetic (oracle/, test infrastructure)
timed on this box's host cores on a bounded row sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# fp32 vector peak and HBM peak of MI355X (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP32_VECTOR_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0
PEAK_ISSUE_SLOTS = 256 * 4 * 32 * 2.4e9  # CUs x SIMDs x lanes/clk x max clock

KERNELS = {"gaussian": "gaussian", "absexp": "absolute-exponential", "invdist": "inverse-distance"}
# per pair: flops counting fma = 2, transcendental = 1 (3D + 2E + 1 for gaussian, D=3, E=1)
# and VALU issue slots counting a quarter-rate transcendental as 4
FLOPS_PER_PAIR = {"gaussian": 12, "inverse-distance": 12, "absolute-exponential": 13}
# VALU issue slots per pair of the pair-loop kernels: lowd_kernel (difference form: 6 for the
# squared distance of the caller's coordinates + 1 multiply by log2 e where the kernel is an
# exponential + transcendental(s) + 1 FMA) and fast_kernel (squared distance on the matrix cores:
# only the transcendental(s) + 1 FMA stay on the VALU)
SLOTS_PER_PAIR = {
    "lowd_kernel": {"gaussian": 12, "inverse-distance": 11, "absolute-exponential": 16},
    "fast_kernel": {"gaussian": 5, "inverse-distance": 5, "absolute-exponential": 9},
    # centred form: + ~0.7 per pair for the operand rebuild and the reach flag
    "cfast_kernel": {"gaussian": 5.7, "inverse-distance": 5.7, "absolute-exponential": 9.7},
    # cell form: the exponential is range-reduced by grid cells and its remainder polynomial comes out of
    # the matrix pipe; per pair the VALU owes ONE fma (the per-cell exponentials amortise to < 0.1 slot)
    "cell_kernel": {"gaussian": 1.0},
}
# VALU flops per pair the cell form needs by construction (one fma); its matrix-pipe flops per pair
CELL_VALU_FLOPS_PER_PAIR = 2
MFMA_FLOPS_PER_PAIR_CELL = 2 * 16  # one 32x32x16 bf16 MFMA per 1024 pairs
MFMA_FLOPS_PER_PAIR_FAST = 2 * 32  # two 32x32x16 bf16 k-steps (K = 6 D + 6 = 24 -> 32) for D = 3
PEAK_BF16_MFMA_TFLOPS = 2500.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--points", dest="n", type=float, default=1e6, help="points (N = M); default = BASELINE config 2")
    p.add_argument("--kernel", choices=sorted(KERNELS), default="gaussian")
    p.add_argument("--precision", choices=["float32", "float64"], default="float32")
    p.add_argument("--sqdists", choices=["auto", "difference", "expanded", "cells"], default="auto",
                   help="squared-distance form (the reference's fast_sqdists flag): auto = expanded form on "
                        "the matrix cores where it is as accurate as the difference form")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--all-ranks-on-device", type=int, default=None,
                   help="debug: put every rank on this one GPU (rehearses the multi-rank path on a 1-GPU box)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time budget of the baseline sample")
    return p.parse_args()


def cpu_baseline(kernel, y64, b64, precision, budget_s):
    """oracle/ (C, OpenMP over target rows) on a bounded sample of target rows."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import c_oracle

    n = y64.shape[0]
    rs = np.random.RandomState(12345)
    threads = c_oracle.threads()
    probe = rs.choice(n, size=min(n, max(64, 2 * threads)), replace=False)
    t0 = time.time()
    c_oracle.product(kernel=kernel, source_points=y64, source_signal=b64, rows=probe, precision=precision)
    dt = max(time.time() - t0, 1e-4)
    rows = int(min(n, max(len(probe), len(probe) * budget_s / dt)))
    sample = rs.choice(n, size=rows, replace=False)
    t0 = time.time()
    c_oracle.product(kernel=kernel, source_points=y64, source_signal=b64, rows=sample, precision=precision)
    dt = time.time() - t0
    return {
        "value": rows * float(n) / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
        "sample": f"{rows} of {n} target rows x all {n} sources, {precision}, oracle/kmvp_oracle.c "
                  f"(OpenMP, {threads} threads), {dt:.1f} s",
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = "WORLD_SIZE" in os.environ
    if world != args.gpus and distributed:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and not distributed:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")

    import numpy as np

    # the host driver of this pool only supports dmabuf IPC (RCCL across processes needs it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # Load the HIP library (system ROCm runtime) BEFORE torch, so that every GPU call of
    # this process goes through one ROCm stack; torch is only used for the gloo
    # rendezvous / barrier / max-over-ranks and never touches the GPU here.
    from kernel_matrix_benchmarks_amd import _lib, sharding
    from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct

    _lib.load()
    dist = None
    comm = None
    if distributed:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        comm = sharding.torch_gloo_communicator()

    def barrier():
        if dist is not None:
            dist.barrier()

    kernel = KERNELS[args.kernel]
    n = int(args.n)
    D, E = 3, 1
    rs = np.random.RandomState(n + D)  # datasets.py:258
    y = rs.rand(n, D)
    b = rs.randn(n, E)

    fast = {"auto": None, "difference": False, "expanded": True, "cells": "cells"}[args.sqdists]
    device = local_rank if args.all_ranks_on_device is None else args.all_ranks_on_device
    algo = MI355XProduct(kernel=kernel, dimension=D, precision=args.precision, device=device, comm=comm,
                         fast_sqdists=fast)
    algo.prepare_data(source_points=y, target_points=y, same_points=True)  # H2D, untimed (runner.py:75-80)
    algo.fit()  # cell order and tile lists (timed by the harness as build_time, not part of a step)
    algo.prepare_query(source_signal=b)
    for _ in range(args.warmup):
        algo.query()

    kernel_ms = []
    barrier()  # query() is synchronous on the device, so the GPU is idle here
    t0 = time.perf_counter()
    for _ in range(args.steps):
        algo.query()  # pair loop + segment reduction + [RCCL all-reduce] + finish, then stream sync
        kernel_ms.append(algo.device_kernel_ms)
    barrier()
    elapsed = time.perf_counter() - t0
    total_ms = algo.device_total_ms
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    a = algo.get_result()
    kname = algo.device_kernel
    # the other squared-distance form on the same resident data, for the record (3 steps, untimed region)
    other = None
    expanded = None
    if args.sqdists == "auto" and kname in ("fast_kernel", "cfast_kernel", "cell_kernel") and world == 1:
        def side_run(code):
            algo.set_query_arguments(fast_sqdists=code)
            algo.query()
            oms = []
            for _ in range(3):
                algo.query()
                oms.append(algo.device_kernel_ms)
            return {"kernel": algo.device_kernel, "kernel_ms": float(np.mean(oms)),
                    "pairs_per_s": float(n) * float(n) / (float(np.mean(oms)) * 1e-3)}

        other = side_run(0)
        if kname == "cell_kernel":
            expanded = side_run(1)  # fast_kernel: one v_exp_f32 per pair, squared distance on the matrix cores
        algo.set_query_arguments(fast_sqdists=-1)
    max_err = rel_err = None
    if rank == 0:
        # max |err| against the float64 oracle on a fixed sample of rows (BASELINE metric's error leg)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import c_oracle

        rows = np.random.RandomState(0).choice(n, size=min(n, 256), replace=False)
        truth = c_oracle.product(kernel=kernel, source_points=y, source_signal=b, rows=rows)
        max_err = float(np.max(np.abs(a[rows] - truth)))
        rel_err = max_err / float(np.max(np.abs(truth)))

    if rank == 0:
        pairs = float(n) * float(n)
        sec_per_step = elapsed / args.steps
        k_ms = float(np.mean(kernel_ms))
        shard_pairs = float(n) * float(algo.shard[1] - algo.shard[0])
        # roofline.achieved counts the flops the bounding unit (VALU) has to execute for this algorithm:
        # SURVEY 8d's 3D + 2E + 1 = 12 per pair for the kernels that evaluate exp() per pair on the VALU;
        # the cell form leaves one fma (2 flops) per pair there -- the rest runs on the matrix pipe, so
        # the 12-flop count is reported beside it as an equivalent, not as a fraction of the vector peak
        flops_per_pair = CELL_VALU_FLOPS_PER_PAIR if kname == "cell_kernel" else FLOPS_PER_PAIR[kernel]
        flops = flops_per_pair * shard_pairs
        achieved_tflops = flops / (k_ms * 1e-3) / 1e12
        traffic = None
        # HBM-side bytes per launch of this exact workload, measured with rocprofv3 PMC passes
        # (tools/profile_bench.sh -> tools/summarize_profile.py); latest committed round wins
        import glob

        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_gaussian_1e6_f32_{kname}_traffic.json")))
        if tfiles and world == 1 and n == 1000000 and kernel == "gaussian" and args.precision == "float32":
            traffic = json.load(open(tfiles[-1])).get("hbm_bytes_per_launch")
        tiles = -(-n // 64)  # one 64-lane wavefront per target tile (T = 1)
        out = {
            "metric": "point-pair interactions/s (N*M/s) + max |err| vs scipy, D=3 Gaussian",
            "value": pairs / sec_per_step,
            "unit": "pairs/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": sec_per_step * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "float32" else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{kernel} product, uniform-3D (uniform_cube seed n+D), N=M={n}, D=3, E=1, "
                            f"{args.precision}, same_points",
                "sharding": f"sources split over {args.gpus} GPU(s), one RCCL all-reduce of (N,E) f64 per step"
                            if args.gpus > 1 else "single GPU",
            },
            "max_abs_err": max_err,
            "max_rel_err": rel_err,
            "roofline": {
                "bound": "valu",
                "kernel": kname,
                "achieved": achieved_tflops,
                "peak": PEAK_FP32_VECTOR_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved_tflops / PEAK_FP32_VECTOR_TFLOPS,
                "traffic": traffic,
                "kernel_ms": k_ms,
                "step_device_ms": total_ms,
                "flops_per_pair": flops_per_pair,
                "survey_equivalent_tflops": FLOPS_PER_PAIR[kernel] * shard_pairs / (k_ms * 1e-3) / 1e12,
                # the same launch priced in VALU issue slots (transcendental = 4 slots)
                "issue_slots_per_pair": SLOTS_PER_PAIR.get(kname, SLOTS_PER_PAIR["lowd_kernel"])[kernel],
                "issue_frac": SLOTS_PER_PAIR.get(kname, SLOTS_PER_PAIR["lowd_kernel"])[kernel] * shard_pairs
                              / (k_ms * 1e-3) / PEAK_ISSUE_SLOTS,
                "mfma_frac": ((MFMA_FLOPS_PER_PAIR_CELL if kname == "cell_kernel" else MFMA_FLOPS_PER_PAIR_FAST)
                              * shard_pairs / (k_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS
                              if kname in ("fast_kernel", "cfast_kernel", "cell_kernel") else 0.0),
                # north-star's "HBM" reading: bytes every wavefront streams from the source block
                "source_stream_GBps": tiles * float(algo.shard[1] - algo.shard[0]) * (D + E) * 4 / (k_ms * 1e-3) / 1e9,
                "source_stream_frac_of_hbm_peak": tiles * float(algo.shard[1] - algo.shard[0]) * (D + E) * 4
                                                  / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                "algorithmic_hbm_bytes": 4 * (n * D + (algo.shard[1] - algo.shard[0]) * (D + E) + n * E),
            },
        }
        out["config"]["sqdists"] = ("cells: exp() range-reduced by grid cells, remainder polynomial of 2 d.e on the bf16 "
                                    "matrix cores (one 32x32x16 MFMA per 1024 pairs), one fma per pair on the VALU"
                                    if kname == "cell_kernel" else
                                    "expanded |x|^2+|y|^2-2x.y on the bf16 matrix cores, 3-way split fp32 operands "
                                    "(reference fast_sqdists=True form)" if kname == "fast_kernel"
                                    else "expanded around per-group centres of Morton-sorted sources on the bf16 matrix "
                                         "cores, closest pairs recomputed exactly" if kname == "cfast_kernel"
                                    else "difference form (reference fast_sqdists=False)")
        if other is not None:
            out["difference_form"] = other
        if expanded is not None:
            out["expanded_form"] = expanded
        if not args.no_cpu_baseline and args.gpus == 1:
            out["cpu_baseline"] = cpu_baseline(kernel, y, b, args.precision, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    algo.done()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
