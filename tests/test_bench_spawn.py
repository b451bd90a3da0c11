"""bench.py --gpus N launches its own ranks: the spawn + rendezvous + bucketed all-gather + max-over-ranks +
single-JSON-line plumbing, exercised at world size 2 on CPU (gloo) with a no-op step (--stub-step).  Also the
refusal when the node has fewer GPUs than asked for."""
import json
import os
import subprocess
import sys

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    return env


def test_self_launch_world2_gloo_stub():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "150", "--warmup", "3", "--stub-step",
                        "--gather-block", "64", "--drones-per-gpu", "1000"], capture_output=True, text=True, timeout=300,
                       env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"rank 0 must print exactly one line, got {lines}"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 150 and out["data"] == "stub" and out["gather_ok"] is True


def test_refuses_more_gpus_than_the_node_has():
    import torch
    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(have + 2), "--steps", "5"], capture_output=True, text=True,
                       timeout=300, env=_clean_env())
    assert r.returncode != 0
    assert f"needs {have + 2} GPUs" in r.stderr and "torch.distributed.run" not in r.stderr


def test_under_a_launcher_the_world_size_must_match():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-step"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
