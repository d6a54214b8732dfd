"""Several ranks on the one GPU of a test box, through the real libkmvp.so (-m gpu).

RCCL cannot form a communicator of two ranks on one device ("invalid usage"), so these runs use the library's
rehearsal transport for the exchange (kmvp_comm_init_host: host-staged, summed by gloo) -- everything else is the
multi-GPU code path: `bench.py --gpus N` starting its own ranks, the plugin's sharding, shard-local kernels with the
global zero rule, the canonical exchange layout, normalisation after the exchange, the sharded solvers.
The box allows at most 6 processes on the card: 1 (pytest) + 3 ranks here.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _spawn(argv_of_rank, world, timeout):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable] + argv_of_rank, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:  # exact PIDs of our own children
            if p.poll() is None:
                p.kill()
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed ({p.returncode}):\n{err[-3000:]}"
    return outs[0][0]


def test_plugin_sharded_over_three_ranks_on_one_gpu():
    out = _spawn([os.path.join(HERE, "_multirank_worker.py")], world=3, timeout=900)
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    rep = json.loads(line)
    assert rep["world"] == 3 and len(rep["cases"]) == 15
    print(json.dumps(rep))


@pytest.mark.parametrize("config,points", [("2", 200000), ("4", 100000)])
def test_bench_starts_its_own_ranks_and_prints_one_line(config, points):
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns the ranks before touching a GPU,
    relays rank 0's JSON line and exits 0.  Both ranks on GPU 0, host-staged exchange."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", config, "--points", str(points),
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--all-ranks-on-device", "0", "--exchange", "host"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["exchange"] == "host"
    assert d["max_rel_err"] <= 2e-5, d["max_rel_err"]
    assert "GPU-side modules loaded in the parent: []" in p.stderr


def test_bench_one_gpu_through_the_self_spawn_path_matches_the_in_process_line():
    """`bench.py --gpus 1 --spawn` (the launcher path with one rank, real RCCL communicator of world 1 is not formed:
    world == 1 attaches nothing) prints the same kind of line as the plain in-process run."""
    common = ["--config", "2", "--points", "200000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    lines = []
    for extra in (["--spawn"], []):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common + extra, env=env,
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        out = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(out) == 1
        lines.append(json.loads(out[0]))
    a, b = lines
    assert a["n_gpus"] == b["n_gpus"] == 1 and a["config"] == b["config"] and a["roofline"]["kernel"] == b["roofline"]["kernel"]
    assert a["max_rel_err"] == b["max_rel_err"]  # same kernels on the same data: bitwise the same answer
