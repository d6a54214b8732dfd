"""What ONE rank of a W-rank run of BASELINE config 2 computes, timed on one GPU (strong scaling rehearsal).

The sources of the uniform cube are split by M over W ranks (bench.py, sharding.py); this tool builds the plugin as rank
W // 2 of W with a rehearsal exchange that adds nothing (the other ranks' sums are simply absent), so the kernel time and
the device step time of that rank's share can be read on a one-GPU box: perfect strong scaling would be T(1) / W.

    python tools/c2_shard.py [--n 1000000] [--worlds 1,2,4,8] [--steps 10]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--n", type=int, default=1000000)
    p.add_argument("--worlds", default="1,2,4,8")
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--kernel", default="gaussian")
    p.add_argument("--shape", type=int, default=-1, help="cellmm_shape option: -1 auto, 0 = 32x32x16, 1 = 16x16x32")
    p.add_argument("--segments", type=int, default=0, help="source segments per launch (0 = the library's choice)")
    a = p.parse_args()
    from kernel_matrix_benchmarks_amd import sharding
    from kernel_matrix_benchmarks_amd.algorithms.mi355x import MI355XProduct

    n, D = a.n, 3
    rs = np.random.RandomState(n + D)
    y = rs.rand(n, D)
    b = rs.randn(n, 1)
    t1 = None
    for w in [int(v) for v in a.worlds.split(",")]:
        comm = None
        if w > 1:
            comm = sharding.Communicator(w // 2, w, lambda payload: payload, host_allreduce=lambda arr, op: None)
        algo = MI355XProduct(kernel=a.kernel, dimension=D, precision="float32", device=0, comm=comm)
        algo.prepare_data(source_points=y, target_points=y, same_points=True)
        algo.set_query_arguments(cellmm_shape=a.shape, segments=a.segments)
        algo.fit()
        algo.prepare_query(source_signal=b)
        for _ in range(5):
            algo.query()
        km, tm = [], []
        for _ in range(a.steps):
            algo.query()
            km.append(algo.device_kernel_ms)
            tm.append(algo.device_total_ms)
        km, tm = float(np.median(km)), float(np.median(tm))
        if t1 is None:
            t1 = (km, tm)
        lo, hi = algo.shard
        print(f"world {w}: rank {w // 2} holds {hi - lo} sources  kernel {algo.device_kernel} {km:8.3f} ms  "
              f"(T1/W = {t1[0] / w:6.3f}; x{t1[0] / km:5.2f})  step on device {tm:8.3f} ms (x{t1[1] / tm:5.2f}; includes the "
              f"host-staged rehearsal exchange for W > 1)", flush=True)
        print("   ", algo.get_additional().get("dispatch_note", ""), flush=True)
        algo.done()


if __name__ == "__main__":
    main()
