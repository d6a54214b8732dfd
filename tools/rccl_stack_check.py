"""Checks that libkmvp.so and the RCCL it dlopens share ONE HIP runtime, in both import
orders (torch first / libkmvp first), with a world-size-1 communicator and a product."""
import os, subprocess, sys
code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
order = sys.argv[1]
if order == "torch_first":
    import torch
from kernel_matrix_benchmarks_amd import _lib
_lib.load()
if order == "kmvp_first":
    import torch
import kmvp_oracle
y, b = kmvp_oracle.uniform_cube(2000, 3)
ctx = _lib.Context(0)
ctx.comm_init(_lib.comm_unique_id(), 0, 1)
ctx.set_points(y.astype(np.float32), None, _lib.KMVP_F32); ctx.set_signal(b.astype(np.float32))
ctx.run("gaussian", True)
got = ctx.get_result(2000, 1)
want = kmvp_oracle.product(kernel="gaussian", source_points=y, source_signal=b, normalize_rows=True)
maps = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "librccl" in l]
print(order, "rel_err %.1e" % (np.max(np.abs(got - want)) / np.max(np.abs(want))), sorted(set(maps)), flush=True)
ctx.close()
'''
for order in ("kmvp_first", "torch_first"):
    r = subprocess.run([sys.executable, "-c", code, order])
    print(order, "exit", r.returncode, flush=True)
