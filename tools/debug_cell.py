"""Debugging aid for the cell kernels: small clouds through cell_kernel / cellmm_kernel against a direct numpy sum."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kernel_matrix_benchmarks_amd import _lib
def run(y, b, x=None, fast=3, tt=1):
    ctx = _lib.Context(0)
    ctx.set_option("fast_sqdists", fast); ctx.set_option("fast_tiles", tt)
    ctx.set_points(y.astype(np.float32), None if x is None else x.astype(np.float32), _lib.KMVP_F32); ctx.set_signal(b.astype(np.float32))
    ctx.run("gaussian", False)
    out = ctx.get_result(len(y) if x is None else len(x), 1); name = ctx.last_kernel_name
    ctx.close(); return out[:, 0], name
def ref(y, b, x=None):
    x = y if x is None else x
    y = y.astype(np.float32).astype(np.float64); x = x.astype(np.float32).astype(np.float64)
    s = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)
    return np.exp(-s) @ b.astype(np.float32).astype(np.float64)[:, 0]
rs = np.random.RandomState(0)
h = np.sqrt(2 * 0.006 / 3)
def report(tag, y, b, x=None):
    got, name = run(y, b, x); want = ref(y, b, x)
    print(tag, name, "max abs err", np.max(np.abs(got - want)), "scale", np.max(np.abs(want)), "got[:4]", got[:4], "want[:4]", want[:4], flush=True)
# 1: 40 points at two opposite corners of a box -> two cells, delta = eps = small
y = np.zeros((40, 3)); y[20:] = 1.0; b = np.ones((40, 1))
report("two corners, b=1:", y, b)
b = rs.randn(40, 1)
report("two corners, random b:", y, b)
# 2: all in one cell, random
y = np.concatenate([rs.rand(38, 3) * h * 0.9, [[0, 0, 0]], [[1, 1, 1]]]); b = rs.randn(40, 1)
report("one cell + far corner:", y, b)
y = rs.rand(500, 3); b = rs.randn(500, 1)
report("uniform 500:", y, b)
report("uniform 500 b=1:", y, np.ones((500, 1)))
def report2(tag, y, b, x=None):
    got, name = run(y, b, x); want = ref(y, b, x)
    print(tag, "direct err", np.max(np.abs(got - want)), "sorted err", np.max(np.abs(np.sort(got) - np.sort(want))), flush=True)
y = rs.rand(500, 3); 
report2("uniform 500 b=1", y, np.ones((500, 1)))
report2("uniform 500 in 2 cells per axis b=1", y * 2 * h * 0.99, np.ones((500, 1)))
report2("uniform 500 in 2 cells per axis", y * 2 * h * 0.99, rs.randn(500, 1))
report2("uniform 500 in 4 cells per axis b=1", y * 4 * h * 0.99, np.ones((500, 1)))
report2("uniform 100 b=1", y[:100], np.ones((100, 1)))
report2("uniform 100 targets vs 40 sources b=1", y[100:140], np.ones((40, 1)), y[:100])
report2("1 target vs 500 sources b=1", y, np.ones((500, 1)), y[:1])
report2("500 targets vs 1 source b=1", y[:1], np.ones((1, 1)), y)
