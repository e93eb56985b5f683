"""bench.py's own launcher on CPU: `--gpus N` without a launcher around it must start N ranks
(one per GPU) by itself, and a world size that contradicts --gpus must fail loudly instead of
silently measuring fewer GPUs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_2_spawns_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_clean_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line == {"launch_check": True, "n_gpus": 2, "requested_gpus": 2}


def test_world_size_that_contradicts_gpus_fails_loudly():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-check"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and "--gpus 4" in r.stderr
