"""cfast_kernel (1, 2, 4 target tiles per wave) against the difference form: time and agreement.  usage: python tools/cmp_cfast.py n [kernel ...]"""
import os, sys
import numpy as np
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/kernel_matrix_benchmarks_amd') else os.environ.get('GRAFT_REPO_ROOT','.'))
from kernel_matrix_benchmarks_amd import _lib
n = int(float(sys.argv[1]))
rs = np.random.RandomState(n + 3)
y = rs.rand(n, 3).astype(np.float32); b = rs.randn(n, 1).astype(np.float32)
for kernel in (sys.argv[2:] or ("inverse-distance", "absolute-exponential", "gaussian")):
    res = {}
    for name, fast, tt in (("lowd", 0, 0), ("cfast1", 2, 1), ("cfast2", 2, 2), ("cfast4", 2, 4)):
        ctx = _lib.Context(0)
        ctx.set_option("fast_sqdists", fast)
        if tt: ctx.set_option("fast_tiles", tt)
        ctx.set_points(y, None, _lib.KMVP_F32); ctx.set_signal(b)
        ctx.run(kernel, False); ctx.run(kernel, False)
        res[name] = (ctx.get_result(n, 1), ctx.last_kernel_ms, ctx.last_kernel_name)
        ctx.close()
    ref = res["lowd"][0]; sc = np.max(np.abs(ref))
    print(kernel, " ".join(f"{k}:{v[2]} {v[1]:.2f}ms err {np.max(np.abs(v[0]-ref))/sc:.2e}" for k, v in res.items()), flush=True)
