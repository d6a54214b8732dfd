#!/usr/bin/env python3
"""Benchmark of the on-the-fly kernel matrix-vector product on MI355X: point-pair interactions per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|4shard|5|attn|softmax]

``--config`` names a BASELINE.json config (default 2, the one the headline metric is quoted on):

  2       Gaussian product, uniform-3D (the reference's ``uniform_cube`` recipe, datasets.py:256-266, seed n+D),
          N = M = 1e6, E = 1, float32.  One step = one product a = K b (1e12 pairs), points and signal resident
          in HBM: what the harness times around ``query()`` (runner.py:138-140).
  3       exp(-r) attention (row-normalised), D = 64, N = M = 65536, E = 64, bf16 MFMA tiles; points scaled by
          1/sqrt(D).  One step = one normalised product.
  4       inverse-distance product, uniform-3D, N = M = 1e7, E = 1, float32, the sources sharded over the ranks
          with ONE RCCL all-reduce of the (N, E) sums per step (BASELINE runs it on 8 GPUs).
  4shard  what one of those 8 ranks computes, on one GPU without a communicator: all 1e7 targets x its 1.25e6
          sources (global zero rule through j_offset / M_total).
  5       Gaussian solver K b = a, N = M = 1e5, D = 3, float64, CG with the HIP matvec as operator, to a
          relative residual of 1e-6.  One step = one solve; pairs = (iterations + 1 products) x N^2.

  2shard  what one of eight ranks computes of config 2 (strong scaling at 8 GPUs), on one GPU: the plugin as rank 4 of 8
          with a rehearsal exchange that adds nothing.  Perfect scaling would be config 2's time / 8.
  softmax (not a BASELINE config) softmax attention exp(<x,y>) (README.md:51-59), row-normalised, x = y ~ N(0,1)^64,
          N = M = 65536, E = 64, bf16 MFMA tiles with the per-target online shift (kmvp_mfma.hpp).

The default run (config 2, one GPU) also measures configs 2shard, 3, softmax, 4shard, 5 and the D = 3 attention shape in the same
process AFTER the timed region of the headline line, a few steps each, and appends them as ``other_configs``
(``--no-other-configs`` skips that).

With N > 1 GPUs the M sources are sharded over the ranks (one process per GPU) and the (N, E) partial sums are
summed by one RCCL all-reduce inside every step: total work is fixed -> "scaling": "strong".  Rank 0 prints ONE
JSON line.  The ranks are either started by ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or, when ``bench.py --gpus N`` is started plainly,
by bench.py itself: the parent process spawns the N ranks BEFORE it imports anything that touches a GPU (it never
loads libkmvp or torch), relays rank 0's line and exits with the worst child status (``launch_ranks``).

``roofline`` describes the dominant kernel from HIP-event timings of every timed launch (events on the library's
own stream, include/kmvp.h ``kmvp_last_kernel_ms``): ``achieved`` = ALGORITHMIC flops per launch (per-pair
figure x pairs, no padding) / average kernel time; ``frac_basis`` says which unit's peak ``peak`` is and why;
``traffic`` is HBM-side bytes per launch from a rocprofv3 PMC run of the same workload (``traffic_source``
names the committed file; it is NOT measured in this run, and null when no profile of this kernel exists).
``cpu_baseline`` is test infrastructure timed on this box's host cores: ``port`` = the C/OpenMP restatement of
the reference's arithmetic (oracle/kmvp_oracle.c) on a bounded row sample of the SAME workload; ``dense`` = the
numpy restatement of what the reference itself does (materialise K in fit(), then ``K @ b`` in query():
bruteforce.py:113-153) at the sizes a dense matrix allows.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md, chip-level parameters
PEAK_F16_MFMA_TFLOPS = 2500.0      # dense bf16 / f16 matrix peak at 2.4 GHz
PEAK_FP32_VECTOR_TFLOPS = 157.3    # 256 CUs x 4 SIMDs x 32 lanes x 2 flop x 2.4 GHz: packed fp32 only
PEAK_FP64_VECTOR_TFLOPS = 78.6
PEAK_HBM_GBPS = 8000.0
# measured on MI355X boxes of this pool (profiles/r02_micro_*.txt), for context next to the data-sheet peaks
SUSTAINED_F16_MFMA_RANDOM_DATA_TFLOPS = 1570.0   # tools/mfma_stream.hip: 1.50-1.61 PFLOP/s on random f16 operands
SUSTAINED_F16_MFMA16_RANDOM_DATA_TFLOPS = 1735.0  # the 16x16x32 shape, two waves per SIMD (profiles/r03_micro_mfma_stream_16x16x32.txt)
NONPACKED_FP32_FMA_TFLOPS = 147.4                # tools/valu_peak.hip: 7.37e13 v_fma_f32 lane-ops/s x 2

KERNELS = {"gaussian": "gaussian", "absexp": "absolute-exponential", "invdist": "inverse-distance"}
FULL_SIZE = {"2": 1000000, "2shard": 1000000, "3": 65536, "4": 10000000, "4shard": 10000000, "5": 100000, "attn": 100000,
             "softmax": 65536}

# Algorithmic work per pair of each pair-loop kernel, on the unit that bounds it.
#   cellmm_kernel: one v_mfma_f32_32x32x16_f16 per 32 x 32 pairs -> 2 x 16 = 32 matrix flop per pair (cellmm16_kernel: the
#                  same on v_mfma_f32_16x16x32_f16, one per 32 sources x 16 targets)
#   cell_kernel:   the same MFMA (bf16) for the polynomial + ONE VALU fma per pair; the VALU is the busy unit
#   fast / cfast:  squared distance on the matrix cores, transcendental + fma per pair on the VALU
#   lowd_kernel:   SURVEY 8d's count, 3 D + 2 E + 1 = 12 flop per pair (fma = 2, exp = 1), all VALU
#   mfma_*_kernel: 2 (D + E + 1) matrix flop per pair (QK^T-like distances + P [b | 1]), SURVEY 8d
#   cell64_kernel: 11 fp64 fma-class instructions per pair = 22 flop
ROOF = {
    "cellmm_kernel": ("mfma", 32.0, PEAK_F16_MFMA_TFLOPS,
                      "32 matrix flop per pair (one 32x32x16 f16 MFMA per 1024 pairs: polynomial remainder of the "
                      "range-reduced exponential, weights and the sum over the sources in the accumulator) vs the dense "
                      "f16 MFMA peak 2.5 PFLOP/s at 2.4 GHz"),
    "cellmm16_kernel": ("mfma", 32.0, PEAK_F16_MFMA_TFLOPS,
                        "32 matrix flop per pair (one 16x16x32 f16 MFMA per 512 pairs -- two groups of 16 sources x 16 monomial "
                        "slots against 16 targets: polynomial remainder of the range-reduced exponential, weights and the sum "
                        "over the sources in the accumulator) vs the dense f16 MFMA peak 2.5 PFLOP/s at 2.4 GHz"),
    "cell_kernel": ("valu", 2.0, PEAK_FP32_VECTOR_TFLOPS,
                    "2 VALU flop per pair (the one fma left after the bf16 MFMA delivers the polynomial) vs the "
                    "packed-fp32 vector peak 157.3 TFLOP/s; v_fma_f32 (not packed) sustains 147 TFLOP/s here"),
    "fast_kernel": ("valu", 12.0, PEAK_FP32_VECTOR_TFLOPS, "SURVEY 8d's 12 flop per pair vs the fp32 vector peak"),
    "cfast_kernel": ("valu", 12.0, PEAK_FP32_VECTOR_TFLOPS, "SURVEY 8d's 12 flop per pair vs the fp32 vector peak"),
    "lowd_kernel": ("valu", 12.0, PEAK_FP32_VECTOR_TFLOPS, "SURVEY 8d's 3D + 2E + 1 = 12 flop per pair vs the fp32 vector peak"),
    "fastmm_kernel": ("valu", 5.0, PEAK_FP32_VECTOR_TFLOPS,
                      "5 VALU flop per pair, whatever the column count (v_exp_f32 = 1, v_cvt_pk_f16_f32 twice = 2, "
                      "v_fma_mix_f32 = 2; both matrix products run on the MFMA pipe beside them) vs the fp32 vector peak; the "
                      "three instructions cost 9.7 / 5.4 / 5.6 issue cycles per wave (profiles/r02_micro_trans_rates_f16.txt), "
                      "i.e. 331 cycles per 32 x 32 tile: see issue_bound_frac"),
    "cell64_kernel": ("valu", 22.0, PEAK_FP64_VECTOR_TFLOPS,
                      "11 fp64 fma-class instructions per pair (3 for t = 2 d.e, 7 Horner steps of exp(t), 1 accumulate) "
                      "= 22 flop vs the fp64 vector peak 78.6 TFLOP/s"),
}

KERNEL_FORM = {
    "cellmm_kernel": "cells: exp() range-reduced by grid cells; polynomial remainder, source weights and the "
                     "sum over the sources in ONE 32x32x16 f16 MFMA per 1024 pairs (fp32 accumulator carried "
                     "over a source cell)",
    "cellmm16_kernel": "cells: exp() range-reduced by grid cells; polynomial remainder, source weights and the "
                       "sum over the sources in ONE 16x16x32 f16 MFMA per 512 pairs (cellmm_kernel's algebra on the MFMA shape "
                       "that sustains more under the power limit; fp32 accumulator carried over a source cell)",
    "cell_kernel": "cells: exp() range-reduced by grid cells, remainder polynomial from one bf16 MFMA per "
                   "1024 pairs, one VALU fma per pair",
    "fast_kernel": "expanded |x|^2+|y|^2-2x.y on the bf16 matrix cores, 3-way split fp32 operands "
                   "(reference fast_sqdists=True form)",
    "cfast_kernel": "expanded around per-group centres of Morton-sorted sources on the bf16 matrix cores, "
                    "closest pairs recomputed exactly",
    "fastmm_kernel": "expanded |x|^2+|y|^2-2x.y on the bf16 matrix cores, exp2 on the VALU, the tile of kernel "
                     "values split into two f16 pieces and multiplied with the (M, E [+1]) signal by a second "
                     "MFMA: up to 32 columns per pass",
    "cell64_kernel": "float64 cells: exp() range-reduced by grid cells, degree-7 remainder on the fp64 VALU",
}


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None, help="default: 10 (config 2, 3), 3 (4shard), 2 (4, 5)")
    p.add_argument("--warmup", type=int, default=None, help="default: 2 (config 2), 50 (the millisecond launches of 3, attn), 1 otherwise")
    p.add_argument("--config", choices=["2", "2shard", "3", "4", "4shard", "5", "attn", "softmax"], default="2",
                   help="BASELINE config; attn (not a BASELINE config): D = 3 Gaussian attention with 16 value channels at "
                        "N = M = 1e5, VERDICT r1 item 9")
    p.add_argument("--points", dest="n", type=float, default=None, help="override N = M of the config (not a BASELINE run)")
    p.add_argument("--kernel", choices=sorted(KERNELS), default=None, help="config 2 only: another kernel function")
    p.add_argument("--precision", choices=["float32", "float64"], default=None, help="config 2 only")
    p.add_argument("--sqdists", choices=["auto", "difference", "expanded", "cells", "cells-valu"], default="auto",
                   help="squared-distance form (the reference's fast_sqdists flag)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-other-configs", action="store_true",
                   help="default run only: do not measure configs 3 / 4shard / 5 / attn after the headline line's timed region")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU time budget of the row-sample baseline")
    p.add_argument("--cpu-dense-sizes", default="10000", help="N = M of the dense numpy baseline (comma separated; "
                                                               "SURVEY 8d asks for 10000,20000,30000)")
    p.add_argument("--all-ranks-on-device", type=int, default=None,
                   help="debug: put every rank on this one GPU (rehearses the multi-rank path on a 1-GPU box; needs "
                        "--exchange host: RCCL refuses two ranks on one device)")
    p.add_argument("--exchange", choices=["rccl", "host"], default="rccl",
                   help="how the ranks' partial sums are summed: rccl (ncclAllReduce on the device, the product) or host "
                        "(rehearsal only: staged through host memory and summed by gloo -- for several ranks on ONE GPU)")
    p.add_argument("--spawn", action="store_true",
                   help="start the ranks from this process even for --gpus 1 (the path a plain `bench.py --gpus N` takes)")
    p.add_argument("--launch-check", action="store_true",
                   help="ranks only rendezvous (gloo) and report their environment; nothing touches a GPU")
    p.add_argument("--rank-timeout", type=float, default=3000.0, help="launcher: seconds before the ranks are stopped")
    return p.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# Launcher: `bench.py --gpus N` started without torch.distributed.run

def launch_ranks(args, argv):
    """Parent of a multi-rank run that was started plainly.  Spawns one child per rank with the environment
    torch.distributed.run would give it (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1,
    MASTER_PORT = a free port), relays rank 0's stdout (the JSON line) to its own stdout and everything else to
    stderr, and returns the worst exit status.  This process never imports torch, ctypes or libkmvp -- no HIP call
    happens here, so the children start from a GPU-clean parent (nothing is exec'ed: they are ordinary children)."""
    import socket
    import threading

    world = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    child_argv = [a for a in argv if a != "--spawn"]
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // world)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + child_argv, env=env,
                                      stdout=subprocess.PIPE, text=True, bufsize=1))
    loaded = [m for m in ("torch", "ctypes", "numpy", "kernel_matrix_benchmarks_amd._lib") if m in sys.modules]
    print(f"[bench launcher] spawned {world} rank(s) on 127.0.0.1:{port}, pids {[p.pid for p in procs]}; "
          f"GPU-side modules loaded in the parent: {loaded}", file=sys.stderr, flush=True)

    def relay(rank, pipe):
        for line in pipe:
            if rank == 0:
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                sys.stderr.write(f"[rank {rank}] {line}")
                sys.stderr.flush()

    threads = [threading.Thread(target=relay, args=(r, p.stdout), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    deadline = time.time() + args.rank_timeout
    worst = 0
    failed_at = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc not in (None, 0) and failed_at is None:
                failed_at = time.time()
                print(f"[bench launcher] rank {r} (pid {p.pid}) exited with status {rc}", file=sys.stderr, flush=True)
        # a rank died: the others would wait for it in a collective for ever -- give them a moment, then stop them
        # (exact PIDs of our own children only)
        if (failed_at is not None and time.time() - failed_at > 15) or time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(5)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            if failed_at is None:
                print("[bench launcher] ranks stopped at --rank-timeout", file=sys.stderr, flush=True)
                worst = 124
            break
    for p in procs:
        p.wait()
    for t in threads:
        t.join(timeout=5)
    for p in procs:
        rc = p.returncode
        if rc != 0:
            worst = max(worst, rc if rc > 0 else 128 - rc)
    return worst


# ---------------------------------------------------------------------------------------------------------
# CPU baselines (test infrastructure: oracle/)

def cpu_exp_dot(y64, b64, budget_s, normalize_rows, np):
    """exp(<x,y>) has no reference plugin (README.md:51-59 only): the CPU figure is the direct numpy evaluation
    (oracle/kmvp_oracle.py exp_dot_product: GEMM + exp + GEMM, float64) on a row sample sized for the budget."""
    import kmvp_oracle

    n = y64.shape[0]
    t0 = time.perf_counter()
    kmvp_oracle.exp_dot_product(source_points=y64, target_points=y64[:256], source_signal=b64, normalize_rows=normalize_rows)
    probe = time.perf_counter() - t0
    rows = int(min(n, max(256, 256 * budget_s / max(probe, 1e-3))))
    t0 = time.perf_counter()
    kmvp_oracle.exp_dot_product(source_points=y64, target_points=y64[:rows], source_signal=b64, normalize_rows=normalize_rows)
    dt = time.perf_counter() - t0
    return {"value": rows * float(n) / dt, "unit": "pairs/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{rows} of {n} target rows x all {n} sources, float64 numpy (GEMM on the BLAS threads, exp on one), {dt:.1f} s"}


def cpu_port(kernel, y64, b64, precision, budget_s, normalize_rows=False, x64=None, shard=None):
    """oracle/kmvp_oracle.c (C, OpenMP over target rows) on a bounded sample of target rows of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import c_oracle

    n = y64.shape[0] if x64 is None else x64.shape[0]
    m = y64.shape[0]
    rs = np.random.RandomState(12345)
    threads = c_oracle.threads()
    kw = dict(kernel=kernel, source_points=y64, source_signal=b64, precision=precision, normalize_rows=normalize_rows)
    if x64 is not None:
        kw["target_points"] = x64
    if shard is not None:
        kw.update(j_offset=shard[0], M_total=shard[1], raw_sums=True)
    probe = rs.choice(n, size=min(n, max(64, 2 * threads)), replace=False)
    t0 = time.time()
    c_oracle.product(rows=probe, **kw)
    dt = max(time.time() - t0, 1e-4)
    rows = int(min(n, max(len(probe), len(probe) * budget_s / dt)))
    sample = rs.choice(n, size=rows, replace=False)
    t0 = time.time()
    c_oracle.product(rows=sample, **kw)
    dt = time.time() - t0
    return {
        "value": rows * float(m) / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
        "sample": f"{rows} of {n} target rows x all {m} sources, {precision}, oracle/kmvp_oracle.c "
                  f"(OpenMP, {threads} threads), {dt:.1f} s",
    }


def cpu_dense(kernel, sizes, D=3):
    """SURVEY 8d's CPU baseline: the numpy restatement of the reference's own algorithm -- fit() materialises the
    N x M kernel matrix (bruteforce.py:113-120, slow (N,M,D) difference form or fast BLAS form), query() is K @ b
    (:150-153) -- on the uniform_cube recipe, float64 and float32, both squared-distance forms.  pairs/s on
    fit + query (what the harness' default axis shows) and on query alone (a GEMV over the stored matrix)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kmvp_oracle

    try:
        from threadpoolctl import threadpool_info

        blas = [f"{i.get('internal_api')} {i.get('version')} x{i.get('num_threads')}" for i in threadpool_info()
                if i.get("user_api") == "blas"]
    except Exception:
        blas = []
    rows = []
    for n in sizes:
        y, b = kmvp_oracle.uniform_cube(n, D)
        for precision in ("float64", "float32"):
            yp, bp = y.astype(precision), b.astype(precision)
            for fast in (False, True):
                t0 = time.time()
                K = kmvp_oracle.kernel_matrix(kernel=kernel, source_points=yp, fast_sqdists=fast)
                fit = time.time() - t0
                t0 = time.time()
                a = K @ bp
                query = time.time() - t0
                del K
                rows.append({"N": n, "precision": precision, "fast_sqdists": fast, "fit_s": round(fit, 4),
                             "query_s": round(query, 5), "pairs_per_s_fit_plus_query": n * float(n) / (fit + query),
                             "pairs_per_s_query_only": n * float(n) / max(query, 1e-9)})
                del a
    return {"kind": "port", "what": "oracle/kmvp_oracle.py kernel_matrix + K @ b (materialise K, then GEMV), "
                                    "uniform_cube, D = 3, E = 1", "host_cores": os.cpu_count(), "blas": blas, "runs": rows}


def head_commit(path):
    """Short hash of the last commit that touched ``path`` (None outside a git checkout)."""
    try:
        out = subprocess.run(["git", "-C", ROOT, "log", "-n1", "--format=%h", "--", path], capture_output=True, text=True,
                             timeout=10)
        return out.stdout.strip() or None
    except Exception:
        return None


def traffic_from_profile(kname, tag):
    """HBM-side bytes per launch from the newest committed rocprofv3 PMC summary of this kernel on this workload."""
    # (the committed summaries of the headline kernel carry cellmm_kernel in their file names for both MFMA shapes)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_{'cellmm_kernel' if kname == 'cellmm16_kernel' else kname}_traffic.json")))
    if not files:
        return None, None
    rel = os.path.relpath(files[-1], ROOT)
    data = json.load(open(files[-1]))
    return data.get("hbm_bytes_per_launch"), {"file": rel, "commit": head_commit(rel) or data.get("library_commit"),
                                              "collected": data.get("collected"), "measured_in_this_run": False,
                                              "note": "rocprofv3 --pmc passes of the same command on another lease; "
                                                      "null when the kernel has no committed profile"}


# ---------------------------------------------------------------------------------------------------------
# Workloads

class Workload:
    """One BASELINE config made resident on the device: ``step()`` is one pass of the hot path (synchronous) and
    returns (pair-loop kernel ms, whole-step device ms); ``result()`` the (N, E) float64 answer of the last step."""

    def __init__(self, cfg, args, device, comm, np, n=None, sqdists="auto", kernel_arg=None, precision_arg=None):
        from kernel_matrix_benchmarks_amd import _lib
        from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct, MI355XSolver

        self.cfg, self.np = cfg, np
        self.D, self.E = 3, 1
        self.normalize = False
        self.solver = False
        self.shard = None
        self.algo = None
        if cfg in ("2", "2shard"):
            self.kernel = KERNELS[kernel_arg or "gaussian"]
            self.precision = precision_arg or "float32"
        elif cfg == "3":
            self.kernel, self.precision, self.normalize = "absolute-exponential", "bfloat16", True
            self.D, self.E = 64, 64
        elif cfg == "attn":
            self.kernel, self.precision, self.normalize = "gaussian", "float32", True
            self.E = 16
        elif cfg == "softmax":
            self.kernel, self.precision, self.normalize = "exp-dot", "bfloat16", True
            self.D, self.E = 64, 64
        elif cfg in ("4", "4shard"):
            self.kernel, self.precision = "inverse-distance", "float32"
        else:
            self.kernel, self.precision, self.solver = "gaussian", "float64", True
        self.n = n = int(n or FULL_SIZE[cfg])
        D, E = self.D, self.E
        rs = np.random.RandomState(n + D)  # datasets.py:258
        y = rs.rand(n, D)
        if cfg == "3":
            y = y / np.sqrt(D)  # otherwise |x - y| ~ 3.3 and every weight is ~ e^-3.3
        if cfg == "softmax":
            y = rs.randn(n, D)  # queries = keys ~ N(0, 1): logits <x, y> ~ N(0, 64), |y|^2/2 spans ~50 (layer-normed heads)
        b = rs.randn(n, E)
        self.y, self.b = y, b
        self.pairs = float(n) * float(n)
        self.a_rhs = None
        self.operator_ms = None
        self.operator_kernel = None
        fast = {"auto": None, "difference": False, "expanded": True, "cells": "cells", "cells-valu": "cells-valu"}[sqdists]

        if cfg == "4shard":
            # one of eight ranks' share, no communicator: the library is told so explicitly (partial_shard)
            r8, w8 = 3, 8
            lo, hi = n * r8 // w8, n * (r8 + 1) // w8
            self.shard = (lo, hi)
            ctx = self.ctx = _lib.Context(device)
            ctx.set_option("same_points_global", 1)
            ctx.set_option("partial_shard", 1)
            y32 = y.astype(np.float32)
            ctx.set_points(np.ascontiguousarray(y32[lo:hi]), y32, _lib.KMVP_F32, j_offset=lo, M_total=n)
            ctx.set_signal(np.ascontiguousarray(b[lo:hi], dtype=np.float32))
            self.pairs = float(n) * float(hi - lo)
        elif self.solver:
            prod = MI355XProduct(kernel=self.kernel, dimension=D, precision=np.float64, device=device)
            prod.prepare_data(source_points=y, target_points=y, same_points=True)
            prod.fit()
            prod.prepare_query(source_signal=b)
            prod.query()
            self.a_rhs = prod.get_result()  # a := K b from the float64 product
            prod.query()                    # (the first query also packs the layouts)
            self.operator_ms = prod.device_kernel_ms
            self.operator_kernel = prod.device_kernel
            prod.done()
            self.algo = MI355XSolver(kernel=self.kernel, dimension=D, precision=np.float64, device=device, rtol=1e-6, maxit=5000)
            self.algo.prepare_data(source_points=y)
            self.algo.fit()
            self.algo.prepare_query(target_signal=self.a_rhs)
            self.device = device
        else:
            if cfg == "2shard":
                # what ONE of eight ranks computes of config 2, on one GPU: the plugin as rank 4 of 8 with a rehearsal exchange
                # that adds nothing (the other ranks' sums are absent) -- kernel choice, cell-wise sharding, canonical exchange
                # layout and the staging of the sums are the real ones, the all-reduce over xGMI is not there
                from kernel_matrix_benchmarks_amd import sharding

                comm = sharding.Communicator(4, 8, lambda payload: payload, host_allreduce=lambda arr, op: None)
            self.algo = MI355XProduct(kernel=self.kernel, dimension=D, normalize_rows=self.normalize, precision=self.precision,
                                      device=device, comm=comm, fast_sqdists=fast)
            self.algo.prepare_data(source_points=y, target_points=y, same_points=True)  # H2D, untimed (runner.py:75-80)
            self.algo.fit()  # cell order and tile lists (the harness books it as build_time; not part of a step)
            self.algo.prepare_query(source_signal=b)
            if cfg == "2shard":
                self.pairs = float(n) * float(self.algo.shard[1] - self.algo.shard[0])

    def step(self):
        if self.cfg == "4shard":
            self.ctx.run(self.kernel, False)
            return self.ctx.last_kernel_ms, self.ctx.last_total_ms
        self.algo.query()  # pair loop + segment reduction + [all-reduce] + finish, then stream sync / one whole solve
        if self.solver:
            return self.operator_ms, 0.0
        return self.algo.device_kernel_ms, self.algo.device_total_ms

    def allreduce_ms(self):
        return 0.0 if self.solver or self.cfg in ("4shard", "2shard") else self.algo._ctx.last_allreduce_ms

    def result(self):
        return self.ctx.get_result(self.n, 1) if self.cfg == "4shard" else self.algo.get_result()

    def info(self):
        if self.cfg == "4shard":
            return {"device_kernel": self.ctx.last_kernel_name, "rccl_ranks": 1, "allreduce_ms": 0.0,
                    "device_bytes": self.ctx.device_bytes}
        d = self.algo.get_additional()
        if self.solver:
            d["device_kernel"] = self.operator_kernel
            d["allreduce_ms"] = 0.0
        return d

    def my_sources(self):
        if self.shard is not None:
            return self.shard[1] - self.shard[0]
        if self.solver:
            return self.n
        return self.algo.shard[1] - self.algo.shard[0]

    def refined_solve(self):
        """The same system by mixed-precision refinement (float64 residuals, float32 CG corrections on the
        matrix-core operator): an extension, reported beside the plain float64 solve, never as `value`."""
        from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XSolver

        np = self.np
        ref = MI355XSolver(kernel=self.kernel, dimension=self.D, precision=np.float64, device=self.device, rtol=1e-6, maxit=5000,
                           refine="float32")
        try:
            ref.prepare_data(source_points=self.y)
            ref.fit()
            ref.prepare_query(target_signal=self.a_rhs)
            ref.query()
            t0 = time.perf_counter()
            ref.query()
            dt = time.perf_counter() - t0
            d = ref.get_additional()
            return {"seconds": dt, "float32_iterations": d["cg_iterations"], "refinement_steps": d["refinement_steps"],
                    "residual_float64": d["cg_relative_residual"], "converged": d["cg_converged"],
                    "inner_kernel": d["inner_device_kernel"], "stop_reason": d.get("refinement_stop_reason")}
        finally:
            ref.done()

    def describe(self, gpus=1, iterations=None):
        n, D, E = self.n, self.D, self.E
        return {
            "2": f"BASELINE config 2: {self.kernel} product, uniform-3D (uniform_cube seed n+D), N=M={n}, D=3, E=1, "
                 f"{self.precision}, same_points",
            "2shard": f"BASELINE config 2, one of 8 source shards on one GPU (rank 4 of 8, sources sharded cell by cell): {n} "
                      f"targets x {self.my_sources() if self.algo is not None else '?'} sources, gaussian, float32; the exchange is a "
                      f"host-staged rehearsal with nothing added",
            "3": f"BASELINE config 3: exp(-r) attention (row-normalised), uniform points / sqrt(D), N=M={n}, D={D}, "
                 f"E={E}, bf16 MFMA tiles, same_points",
            "attn": f"not a BASELINE config (VERDICT r1 item 9): Gaussian attention (row-normalised), uniform-3D, "
                    f"N=M={n}, D=3, E={E} value channels, float32, same_points",
            "softmax": f"not a BASELINE config (SURVEY 8f-4, README.md:51-59): softmax attention exp(<x,y>) (row-normalised), "
                       f"x = y ~ N(0, 1)^D, N=M={n}, D={D}, E={E}, bf16 MFMA tiles with the per-target online shift",
            "4": f"BASELINE config 4: inverse-distance product, uniform-3D, N=M={n}, D=3, E=1, float32, sources "
                 f"sharded over {gpus} GPU(s)",
            "4shard": f"BASELINE config 4, one of 8 source shards on one GPU: {n} targets x {self.my_sources()} sources "
                      f"(rank 3 of 8, global zero rule), inverse-distance, float32",
            "5": f"BASELINE config 5: Gaussian solver K b = a, uniform-3D, N=M={n}, D=3, float64, CG on the HIP "
                 f"matvec, rtol 1e-6; one step = one solve ({iterations} iterations + 1 residual product)",
        }[self.cfg]

    def done(self):
        if self.cfg == "4shard":
            self.ctx.close()
        else:
            self.algo.done()


def time_steps(W, steps, warmup, barrier):
    """W untimed warm-up steps, then EXACTLY `steps` steps between two barriers (every step is synchronous on the
    device, so the GPU is idle at both)."""
    for _ in range(warmup):
        W.step()
    kernel_ms, total_ms, allreduce_ms = [], [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        k_ms, t_ms = W.step()
        kernel_ms.append(k_ms)
        total_ms.append(t_ms)
        allreduce_ms.append(W.allreduce_ms())
    barrier()
    return time.perf_counter() - t0, kernel_ms, total_ms, allreduce_ms


def error_leg(W, a, meta):
    """Against the float64 C restatement of the reference on 256 sampled rows (solver: the residual on those rows)."""
    import c_oracle

    np = W.np
    rows = np.random.RandomState(0).choice(W.n, size=min(W.n, 256), replace=False)
    if W.solver:
        Kb = c_oracle.product(kernel=W.kernel, source_points=W.y, source_signal=a, rows=rows)
        return None, None, {"residual_rows": float(np.linalg.norm(Kb - W.a_rhs[rows]) / np.linalg.norm(W.a_rhs[rows])),
                            "residual_reported": float(meta["cg_relative_residual"]), "iterations": int(meta["cg_iterations"]),
                            "converged": bool(meta["cg_converged"])}
    if W.shard is not None:
        lo, hi = W.shard
        truth, _ = c_oracle.product(kernel=W.kernel, source_points=W.y[lo:hi], target_points=W.y, source_signal=W.b[lo:hi],
                                    rows=rows, j_offset=lo, M_total=W.n, raw_sums=True)
    elif W.cfg == "2shard":
        lo, hi = W.algo.shard
        order = W.algo._order if W.algo._order is not None else np.arange(W.n)
        truth = c_oracle.product(kernel=W.kernel, source_points=W.y[order][lo:hi], target_points=W.y, source_signal=W.b[order][lo:hi],
                                 rows=rows)
    elif W.kernel == "exp-dot":
        import kmvp_oracle  # (the C restatement has the reference's three kernels; exp(<x,y>) is checked by direct evaluation)

        truth = kmvp_oracle.exp_dot_product(source_points=W.y, target_points=W.y[rows], source_signal=W.b, normalize_rows=W.normalize)
    else:
        truth = c_oracle.product(kernel=W.kernel, source_points=W.y, source_signal=W.b, rows=rows, normalize_rows=W.normalize)
    norms = np.sqrt(np.sum((a[rows] - truth) ** 2, axis=-1))  # plotting/metrics.py:53-56
    max_err = float(np.max(norms))
    return max_err, max_err / float(np.max(np.sqrt(np.sum(truth ** 2, axis=-1)))), {}


def roofline_of(W, kname, k_ms, world=1):
    """bound / achieved / peak / frac of the pair-loop kernel `kname` at `k_ms` per launch on workload W."""
    n, D, E = W.n, W.D, W.E
    my_sources = W.my_sources()
    shard_pairs = float(n) * float(my_sources)
    if kname.startswith("mfma"):
        bound, fpp, peak, basis = ("mfma", 2.0 * (D + E + 1), PEAK_F16_MFMA_TFLOPS,
                                   "2 (D + E + 1) matrix flop per pair (distances + P [b | 1], SURVEY 8d) vs the dense "
                                   "bf16 MFMA peak; the transcendental rate bounds it equally (" +
                                   ("one sqrt + one exp2" if W.kernel == "absolute-exponential" else "one exp2") + " per pair)")
    else:
        bound, fpp, peak, basis = ROOF.get(kname, ROOF["lowd_kernel"])
    achieved = fpp * shard_pairs / (k_ms * 1e-3) / 1e12
    cfg = W.cfg
    tag = {"2": f"{'gaussian' if W.kernel == 'gaussian' else W.kernel}_1e6_{'f32' if W.precision == 'float32' else 'f64'}",
           "2shard": "c2shard_gaussian_f32",
           "3": "c3_absexp_bf16", "4": "c4_invdist_1e7_f32", "4shard": "c4shard_invdist_f32", "5": "c5_gaussian_1e5_f64",
           "attn": "attn_gaussian_1e5_e16_f32", "softmax": "softmax_expdot_bf16"}[cfg]
    traffic, traffic_source = traffic_from_profile(kname, tag) if n == FULL_SIZE[cfg] and world == 1 else (None, None)
    esize = 8 if W.precision == "float64" else (2 if W.precision == "bfloat16" else 4)
    r = {
        "bound": bound,
        "kernel": kname,
        "achieved": achieved,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": achieved / peak,
        "frac_basis": basis,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "kernel_ms": k_ms,
        "flops_per_pair": fpp,
        "algorithmic_hbm_bytes": esize * (n * D + my_sources * (D + E) + n * E),
    }
    if bound == "mfma" and kname in ("cellmm_kernel", "cellmm16_kernel"):
        sustained = SUSTAINED_F16_MFMA16_RANDOM_DATA_TFLOPS if kname == "cellmm16_kernel" else SUSTAINED_F16_MFMA_RANDOM_DATA_TFLOPS
        r["sustained_peak_random_data"] = sustained
        r["frac_of_sustained"] = achieved / sustained
        r["sustained_note"] = ("tools/mfma_stream.hip on this pool: the same MFMA sustains 2.48 PFLOP/s on zero operands "
                               "but 1.50-1.61 PFLOP/s on random f16 operands (power limit, clock ~2.0 GHz), the 16x16x32 "
                               "shape 1.74 PFLOP/s at two waves per SIMD: profiles/r02_micro_mfma_stream.txt, "
                               "profiles/r03_micro_mfma_stream_16x16x32.txt")
    if kname in ("cellmm_kernel", "cellmm16_kernel", "cell_kernel", "fast_kernel", "cfast_kernel") and cfg in ("2", "4", "4shard"):
        # SURVEY 8d's VALU model (12 flop per pair against the fp32 vector peak) does not describe kernels whose
        # exponential / squared distance runs on the matrix pipe; reported as an equivalent only
        r["survey_equivalent_tflops"] = 12.0 * shard_pairs / (k_ms * 1e-3) / 1e12
        r["survey_equivalent_note"] = ("SURVEY 8d prices a pair at 12 VALU flop; this kernel moved that work to the matrix "
                                       "pipe, so the figure may exceed the 157.3 TFLOP/s vector peak: it is an "
                                       "equivalent, not a fraction of any unit's peak")
    if kname == "fastmm_kernel":
        # VALU issue bound at the measured instruction costs, at the nominal 2.4 GHz
        tiles32 = -(-n // 32) * -(-my_sources // 32) * (-(-(E + (1 if W.normalize else 0)) // 32))
        r["issue_bound_ms"] = tiles32 * 331.0 / 1024 / 2.4e9 * 1e3
        r["issue_bound_frac"] = r["issue_bound_ms"] / k_ms
    if kname == "cfast_kernel":
        # per 32 x 32 tile of pairs a wave issues 16 transcendentals (9.76 cycles each with >= 4 waves per SIMD,
        # profiles/r02_micro_trans_rates_f16.txt), 16 v_fmac_f32 (3.35) and 2 MFMAs (8 issue cycles); packing the FMAs
        # into v_pk_fma_f32 (5.68 per two) changed nothing measurable (LAB_NOTES I.8)
        per_tile = (32 if kernel_uses_two_transcendentals(W.kernel) else 16) * 9.76 + 16 * 3.35 + 2 * 8
        r["issue_bound_ms"] = (-(-n // 32)) * (-(-my_sources // 32)) * per_tile / 1024 / 2.4e9 * 1e3
        r["issue_bound_frac"] = r["issue_bound_ms"] / k_ms
    if kname == "mfma_pipe_kernel" and W.kernel in ("absolute-exponential", "exp-dot"):
        # 16 v_sqrt + 16 v_exp (9.76; exp(<x,y>): the 16 v_exp only), 16 v_add for the denominators (3.35), 8 v_cvt_pk_bf16
        # (5.42) and the issue of KS + 2 NT MFMAs (8) per 32 x 32 tile of pairs
        ks, nt = -(-(D + (6 if W.kernel == "absolute-exponential" else 3)) // 16), -(-E // 32)
        per_tile = (32 if W.kernel == "absolute-exponential" else 16) * 9.76 + 16 * 3.35 + 8 * 5.42 + (ks + 2 * nt) * 8
        r["issue_bound_ms"] = (-(-n // 32)) * (-(-my_sources // 32)) * per_tile / 1024 / 2.4e9 * 1e3
        r["issue_bound_frac"] = r["issue_bound_ms"] / k_ms
    if kname == "cell_kernel":
        r["nonpacked_fp32_ceiling_tflops"] = NONPACKED_FP32_FMA_TFLOPS
        r["mfma_frac"] = 32.0 * shard_pairs / (k_ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS
    if cfg in ("2", "4", "4shard"):
        tiles = -(-n // 64)  # the north star's reading: every 64-target wavefront streams the whole source block
        r["source_stream_GBps"] = tiles * float(my_sources) * (D + E) * 4 / (k_ms * 1e-3) / 1e9
        r["source_stream_frac_of_hbm_peak"] = r["source_stream_GBps"] / PEAK_HBM_GBPS
    return r


def kernel_uses_two_transcendentals(kernel):
    """exp(-r) takes a square root and an exponential per pair; the Gaussian and 1/r take one instruction."""
    return kernel == "absolute-exponential"


def measure_other_configs(args, device, np):
    """Configs 3, 4shard, 5 and the D = 3 attention shape, a few steps each, in this process, after the headline
    line's timed region: so that the driver's own record carries every BASELINE config, not only config 2."""
    # (the short launches need a long warm-up: after the idle seconds of the CPU baseline the chip takes ~50 ms of work to
    # return to its steady clock -- config 3: 1.30 ms per launch after 2 warm-up steps, 1.18 after 40, 1.10-1.12 in steady
    # state; profiles/r03_c3_variants.txt)
    plan = (("2shard", 10, 10), ("3", 20, 50), ("softmax", 20, 50), ("attn", 20, 50), ("4shard", 2, 1), ("5", 1, 1))
    out = {}
    for cfg, steps, warmup in plan:
        t_start = time.time()
        try:
            W = Workload(cfg, args, device, None, np)
            try:
                elapsed, kernel_ms, total_ms, _ = time_steps(W, steps, warmup, lambda: None)
                a = W.result()
                meta = W.info()
                kname = meta["device_kernel"]
                max_err, rel_err, err = error_leg(W, a, meta)
                k_ms = float(np.mean(kernel_ms))
                pairs_per_step = W.pairs * (meta["cg_iterations"] + 1) if W.solver else W.pairs
                r = roofline_of(W, kname, k_ms)
                entry = {
                    "workload": W.describe(iterations=err.get("iterations")),
                    "dtype": {"float32": "f32", "float64": "f64", "bfloat16": "bf16"}[W.precision],
                    "steps": steps, "warmup": warmup,
                    "kernel": kname, "kernel_ms": k_ms, "ms_per_step": elapsed / steps * 1e3,
                    "pairs_per_s": pairs_per_step / (elapsed / steps),
                    "pairs_per_s_kernel": W.pairs / (k_ms * 1e-3),
                    "max_abs_err": max_err, "max_rel_err": rel_err,
                    "roofline": {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_basis", "traffic",
                                                   "flops_per_pair", "algorithmic_hbm_bytes") if k in r},
                }
                if "issue_bound_frac" in r:
                    entry["roofline"]["issue_bound_frac"] = r["issue_bound_frac"]
                if W.solver:
                    entry["solver"] = err
                    entry["operator_ms"] = W.operator_ms
                note = meta.get("dispatch_note")
                if note:
                    entry["dispatch_note"] = note
            finally:
                W.done()
                del W
        except Exception as e:  # the headline line must still be printed
            entry = {"error": f"{type(e).__name__}: {e}"}
        entry["wall_s"] = round(time.time() - t_start, 2)
        out["C" + cfg if cfg not in ("attn", "softmax") else cfg] = entry
    return out


# ---------------------------------------------------------------------------------------------------------

def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    distributed = "WORLD_SIZE" in os.environ
    if not distributed and (args.gpus > 1 or args.spawn):
        # started plainly: this process becomes the launcher of the N ranks; it must not touch a GPU
        sys.exit(launch_ranks(args, argv))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # every rank checks the launch shape BEFORE anything touches a GPU, with a message that says what to change
    if distributed and world != args.gpus:
        raise SystemExit(f"[rank {rank}] --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if args.config in ("2shard", "4shard", "5", "3", "attn", "softmax") and args.gpus > 1:
        raise SystemExit(f"--config {args.config} is a single-GPU measurement")
    # the host driver of this pool only supports dmabuf IPC (RCCL across processes needs it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    if args.launch_check:
        # rendezvous only: proves the launch shape (environment, gloo group of `world` ranks) without a GPU
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.tensor([rank], dtype=torch.int64)
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"launch_check": True, "world": dist.get_world_size(), "sum_of_ranks": int(t[0]),
                              "rank0_env": {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                                                                           "MASTER_ADDR", "MASTER_PORT")}}), flush=True)
        dist.destroy_process_group()
        return

    import numpy as np

    # Load the HIP library (system ROCm runtime) BEFORE torch, so that every GPU call of this process goes
    # through one ROCm stack; torch is only used for the gloo rendezvous / barrier / max-over-ranks.
    from kernel_matrix_benchmarks_amd import _lib, sharding

    _lib.load()
    device = local_rank if args.all_ranks_on_device is None else args.all_ranks_on_device
    visible = _lib.device_count()  # hipGetDeviceCount only: no context is created
    if visible < device + 1:
        raise SystemExit(f"[rank {rank}] LOCAL_RANK={local_rank} needs GPU {device} but this process sees {visible} "
                         f"device(s) (HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')}, "
                         f"ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')})")
    dist = None
    comm = None
    if distributed:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        comm = sharding.torch_gloo_communicator(exchange=args.exchange)

    def barrier():
        if dist is not None:
            dist.barrier()

    cfg = args.config
    steps = args.steps if args.steps is not None else {"2": 10, "2shard": 10, "3": 10, "4shard": 3, "4": 2, "5": 2, "attn": 10, "softmax": 10}[cfg]
    warmup = args.warmup if args.warmup is not None else {"2": 2, "2shard": 10, "3": 50, "attn": 50, "softmax": 50}.get(cfg, 1)

    W = Workload(cfg, args, device, comm, np, n=args.n, sqdists=args.sqdists, kernel_arg=args.kernel,
                 precision_arg=args.precision)
    n, D, E, kernel, precision, solver = W.n, W.D, W.E, W.kernel, W.precision, W.solver
    elapsed, kernel_ms, total_ms, allreduce_ms = time_steps(W, steps, warmup, barrier)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    a = W.result()
    meta = W.info()
    kname = meta["device_kernel"]
    rccl_ranks = int(meta.get("rccl_ranks", 1))
    if dist is not None:
        # every rank's communicator must span all ranks: a line with n_gpus = N and fewer exchanging ranks is void
        seen = torch.tensor([rccl_ranks], dtype=torch.int64)
        dist.all_reduce(seen, op=dist.ReduceOp.MIN)
        rccl_ranks = int(seen[0])
        if rccl_ranks != world and cfg != "2shard":
            raise SystemExit(f"[rank {rank}] the communicator spans {rccl_ranks} rank(s), expected {world}")

    # the other forms on the same resident data, for the record (config 2, one GPU, untimed region)
    others = {}
    if cfg == "2" and args.sqdists == "auto" and world == 1 and kname in ("cellmm_kernel", "cellmm16_kernel", "cell_kernel", "fast_kernel"):
        algo = W.algo

        def side_run(code):
            algo.set_query_arguments(fast_sqdists=code)
            algo.query()
            oms = []
            for _ in range(3):
                algo.query()
                oms.append(algo.device_kernel_ms)
            return {"kernel": algo.device_kernel, "kernel_ms": float(np.mean(oms)),
                    "pairs_per_s": W.pairs / (float(np.mean(oms)) * 1e-3)}

        others["difference_form"] = side_run(0)
        others["expanded_form"] = side_run(1)
        if kname in ("cellmm_kernel", "cellmm16_kernel"):
            others["cell_form_valu_sum"] = side_run(4)
        algo.set_query_arguments(fast_sqdists=-1)

    out = None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        # ---- error leg of the metric: against the float64 C restatement of the reference on sampled rows
        max_err, rel_err, err = error_leg(W, a, meta)
        pairs_per_step = W.pairs * (meta["cg_iterations"] + 1) if solver else W.pairs  # + the true-residual product

        sec_per_step = elapsed / steps
        k_ms = float(np.mean(kernel_ms))
        r = roofline_of(W, kname, k_ms, world)
        r["step_device_ms"] = float(np.mean(total_ms)) if not solver else None
        r["allreduce_ms"] = float(np.mean(allreduce_ms)) if allreduce_ms else 0.0
        out = {
            "metric": "point-pair interactions/s (N*M/s) + max |err| vs scipy, D=3 Gaussian",
            "value": pairs_per_step / sec_per_step,
            "unit": "pairs/s",
            "n_gpus": args.gpus,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": sec_per_step * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": {"float32": "f32", "float64": "f64", "bfloat16": "bf16"}[precision],
            "data": "synthetic",
            "config": {
                "workload": W.describe(args.gpus, err.get("iterations")),
                "sharding": (f"sources split over {args.gpus} GPUs, one "
                             f"{'RCCL all-reduce' if args.exchange == 'rccl' else 'HOST-STAGED gloo all-reduce (rehearsal)'} of "
                             f"the (N,E) f64 sums per step" if args.gpus > 1 else "single GPU"),
                "kernel_form": KERNEL_FORM.get(kname, "difference form (reference fast_sqdists=False)" if "lowd" in kname else kname),
            },
            "rccl_ranks": rccl_ranks,
            "exchange": ("host-staged rehearsal as rank 4 of 8 with nothing added: rccl_ranks is that communicator's size"
                         if cfg == "2shard" else (args.exchange if args.gpus > 1 else None)),
            "device_bytes": meta.get("device_bytes"),
            "max_abs_err": max_err,
            "max_rel_err": rel_err,
            "error_reference": ("direct float64 numpy evaluation of exp(<x,y>) (oracle/kmvp_oracle.py exp_dot_product; PARITY "
                                "UNPINNED: the kernel exists in the reference's README only), 256 sampled rows"
                                if kernel == "exp-dot" else
                                "float64 oracle/kmvp_oracle.c (C restatement of the reference's scipy/numpy bruteforce, pinned "
                                "by tests/golden), 256 sampled rows; all rows: tests/test_gpu_parity.py::test_config2_*"),
            "roofline": r,
        }
        if solver:
            out["solver"] = err
            out["config"]["operator_ms"] = W.operator_ms
            out["solver"]["mixed_precision_refinement"] = W.refined_solve()
        out.update(others)
        if not args.no_cpu_baseline and args.gpus == 1:
            if W.shard is not None:
                lo, hi = W.shard
                base = cpu_port(kernel, W.y[lo:hi], W.b[lo:hi], "float32", args.cpu_seconds, x64=W.y, shard=(lo, n))
            elif kernel == "exp-dot":
                base = cpu_exp_dot(W.y, W.b, args.cpu_seconds, W.normalize, np)
            else:
                base = cpu_port(kernel, W.y, W.b, "float32" if precision == "bfloat16" else precision, args.cpu_seconds,
                                normalize_rows=W.normalize)
            if solver:
                base["sample"] += " -- ONE operator application; the reference's dense lstsq (bruteforce.py:193-207) needs " \
                                  "80 GB and O(M^3) = 1e15 flop at this size and cannot run"
            sizes = [int(float(s)) for s in args.cpu_dense_sizes.split(",") if s]
            if sizes and cfg == "2":
                base["dense"] = cpu_dense(kernel, sizes)
            out["cpu_baseline"] = base
    W.done()
    default_shape = (cfg == "2" and world == 1 and n == FULL_SIZE["2"] and args.kernel is None and args.precision is None
                     and args.sqdists == "auto")
    if rank == 0:
        if default_shape and not args.no_other_configs:
            del W, a
            out["other_configs"] = measure_other_configs(args, device, np)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
