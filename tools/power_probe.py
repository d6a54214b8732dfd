"""Runs one kernel form in a loop for ~8 s while sampling rocm-smi (power, sclk) -- is the launch power-limited?
usage: python tools/power_probe.py <fast_sqdists code> [n]"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kernel_matrix_benchmarks_amd import _lib

code = int(sys.argv[1]); n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
rs = np.random.RandomState(n + 3)
y = rs.rand(n, 3).astype(np.float32); b = rs.randn(n, 1).astype(np.float32)
if len(sys.argv) > 3 and sys.argv[3] == "zero":
    b[:] = 0  # zero operands: how much of the kernel's time is data-dependent (power)?
if len(sys.argv) > 3 and sys.argv[3] == "ones":
    b[:] = 1
ctx = _lib.Context(0)
ctx.set_option("fast_sqdists", code)
ctx.set_points(y, None, _lib.KMVP_F32); ctx.set_signal(b)
ctx.run("gaussian", False)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out.strip().splitlines()[-1] if out.strip() else "")
        except Exception as e:
            samples.append(str(e))
        time.sleep(0.3)
t = threading.Thread(target=sampler); t.start()
t0 = time.time(); ms = []
while time.time() - t0 < 4:
    ctx.run("gaussian", False); ms.append(ctx.last_kernel_ms)
stop = True; t.join()
print(ctx.last_kernel_name, "kernel ms: first %.3f  min %.3f  median %.3f  last %.3f  (%d launches)" % (ms[0], min(ms), float(np.median(ms)), ms[-1], len(ms)))
hdr = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()
print(hdr[0] if hdr else "")
for s in samples[:: max(1, len(samples) // 8)]:
    print(s)
ctx.close()
