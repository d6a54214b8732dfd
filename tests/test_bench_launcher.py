"""bench.py as its own launcher (CPU): `python bench.py --gpus N` started plainly must start its N ranks itself,
from a parent that has loaded nothing GPU-side, give them the environment torch.distributed.run would, relay rank 0's
single JSON line and return the worst child status."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                                                             "MASTER_ADDR", "MASTER_PORT")}


def test_plain_start_with_two_gpus_spawns_two_ranks_that_rendezvous():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_clean_env(), capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE line, rank 0's
    d = json.loads(lines[0])
    assert d["launch_check"] and d["world"] == 2 and d["sum_of_ranks"] == 1
    env0 = d["rank0_env"]
    assert env0["RANK"] == "0" and env0["LOCAL_RANK"] == "0" and env0["WORLD_SIZE"] == "2" and env0["MASTER_ADDR"] == "127.0.0.1"
    assert int(env0["MASTER_PORT"]) > 0
    # the parent spawned before importing anything that could initialise HIP
    assert "GPU-side modules loaded in the parent: []" in p.stderr, p.stderr[-2000:]


def test_a_failing_rank_fails_the_launcher_and_stops_the_others():
    # WORLD_SIZE mismatch inside the children (--gpus 3 handed to ranks that are told WORLD_SIZE=3 is fine; force a
    # failure instead through an impossible config combination that every rank rejects before any GPU call)
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--config", "5"], env=_clean_env(), capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0
    assert "single-GPU measurement" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_under_a_launcher_the_world_size_must_match():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-check"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert p.returncode != 0 and "--nproc-per-node 4" in p.stderr
