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
    # the self-certifying part of an N-rank line: what the process group itself reports
    c = out["collective"]
    assert c["world_seen"] == 2 and c["world_env"] == 2 and c["backend"] == "gloo" and c["rank_ids_gathered"] == [0, 1]
    assert len(c["per_rank_ms_per_step"]["all"]) == 2 and c["per_rank_ms_per_step"]["min"] <= c["per_rank_ms_per_step"]["max"]
    words = (1000 + 63) // 64
    assert c["gather"]["block_steps"] == 64 and c["gather"]["bytes_per_bucket"] == 64 * words * 8
    assert c["gather"]["collectives_launched"] == 150 // 64 + 1          # two whole buckets + the flushed tail
    assert c["library_version"]


def test_a_rank_that_dies_early_ends_the_job_at_once():
    """rank 1 exits 1 before init_process_group; rank 0 would wait in the rendezvous for the backend's own
    timeout (10-30 min).  The parent polls every child, stops the survivors and returns the failing code - quickly."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "50", "--stub-step", "--stub-fail-rank", "1",
                        "--drones-per-gpu", "1000"], capture_output=True, text=True, timeout=300, env=_clean_env())
    took = time.monotonic() - t0
    assert r.returncode == 1, (r.returncode, r.stderr[-2000:])
    assert "rank 1 exited with 1" in r.stderr and took < 120, took
    assert not [ln for ln in r.stdout.splitlines() if ln.strip()], "no JSON line from a failed job"


def test_overall_wall_clock_limit_of_self_launched_ranks():
    """Both ranks alive but stuck (rank 1 never joins: it fails only after a sleep is impossible to stage here, so the
    limit itself is exercised with a job that cannot finish in time): exit code 124 and no orphan processes."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2000000", "--stub-step", "--gather-block", "1",
                        "--drones-per-gpu", "100000", "--spawn-timeout-s", "6"], capture_output=True, text=True, timeout=300,
                       env=_clean_env())
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert "still running after 6 s" in r.stderr


def _run_stub(extra, env=None, timeout=300):
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "40", "--warmup", "2", "--stub-step", "--drones-per-gpu", "1000"] + extra,
                       capture_output=True, text=True, timeout=timeout, env=env or _clean_env())
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    return r, lines, time.monotonic() - t0


def test_preflight_failure_falls_back_to_gloo_and_says_so():
    """RCCL that does not come up must not cost the line.  The preflight children fail (with the
    caller's IPC mode, then with the other one); fresh workers run with the done mask over gloo, and the line says what
    happened."""
    r, lines, _ = _run_stub(["--stub-preflight", "fail"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    c = json.loads(lines[0])["collective"]
    assert c["requested"] == "rccl" and c["used"] == "gloo" and c["backend"] == "gloo"
    assert "preflight failed" in c["fallback_reason"] and "rank 0" in c["fallback_reason"]
    assert [p["ok"] for p in c["preflight"]] == [False, False] and [p["ipc_mode"] for p in c["preflight"]] == ["0", "1"]
    assert [a["collective"] for a in c["attempts"]] == ["gloo"] and c["attempts"][0]["ok"]
    assert c["world_seen"] == 2 and c["rank_ids_gathered"] == [0, 1] and c["gather"]["collectives_launched"] >= 1
    assert c["done_mask_exchange"].startswith("gloo")


def test_preflight_that_hangs_is_stopped_at_its_limit():
    r, lines, took = _run_stub(["--stub-preflight", "hang", "--preflight-timeout-s", "3"])
    assert r.returncode == 0, r.stderr[-3000:]
    c = json.loads(lines[0])["collective"]
    assert c["used"] == "gloo" and "no answer within 3 s" in c["fallback_reason"]
    assert took < 120, took


def test_callers_ipc_mode_is_honoured_and_recorded():
    """bench.py used to hard-set HSA_ENABLE_IPC_MODE_LEGACY=0 in every child; now a caller's value travels to the
    preflight and the workers and is written into the line; without one the default is 0 and the line says it is ours."""
    env = dict(_clean_env(), HSA_ENABLE_IPC_MODE_LEGACY="1")
    r, lines, _ = _run_stub([], env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    c = json.loads(lines[0])["collective"]
    assert c["ipc_mode"] == "1" and c["ipc_mode_of_caller"] == "1" and c["ipc_mode_source"] == "caller's environment"
    assert c["used"] == "rccl" and c["fallback_reason"] is None and c["preflight"][0]["ok"] and c["preflight"][0]["ipc_mode"] == "1"
    env = _clean_env()
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    r, lines, _ = _run_stub([], env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    c = json.loads(lines[0])["collective"]
    assert c["ipc_mode"] == "0" and c["ipc_mode_source"] == "bench.py default"


def test_a_run_that_fails_in_the_collective_is_repeated_with_the_fallback():
    r, lines, _ = _run_stub(["--stub-fail-collective", "rccl"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, "only the accepted attempt's line is printed"
    c = json.loads(lines[0])["collective"]
    assert [(a["collective"], a["ok"]) for a in c["attempts"]] == [("rccl", False), ("gloo", True)]
    assert c["used"] == "gloo" and c["fallback_reason"].startswith("run failed") and "exit code 1" in c["fallback_reason"]
    assert "the run with --collective rccl failed" in r.stderr


def test_a_run_that_hangs_in_the_collective_is_stopped_and_repeated_with_the_fallback():
    """The likeliest way for RCCL to fail on an unknown node is not an error but a hang: one rank never comes back from
    the collective.  Its supervisor stops it at the attempt limit (the peers' workers leave through the process-group
    timeout or are stopped too), all ranks agree that the attempt failed, and fresh workers run over gloo."""
    r, lines, took = _run_stub(["--stub-hang-collective", "rccl", "--attempt-timeout-s", "8", "--pg-timeout-s", "6"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    c = json.loads(lines[0])["collective"]
    assert [(a["collective"], a["ok"]) for a in c["attempts"]] == [("rccl", False), ("gloo", True)]
    assert c["used"] == "gloo" and "no result within 8 s" in c["fallback_reason"]
    assert took < 150, took
    import time
    left = []
    for _ in range(20):                 # a stopped worker may take a moment to be reaped; none may survive
        left = [ln for ln in subprocess.run(["pgrep", "-af", "stub-hang-collective rccl"], capture_output=True, text=True).stdout.splitlines()
                if "pgrep" not in ln]
        if not left:
            break
        time.sleep(0.5)
    assert not left, f"workers left behind: {left}"


def test_stage_limits_share_one_deadline():
    """bench.stage_limit: a stage gets min(its configured limit, time left - reserve x attempts still in the plan - margin);
    under 5 s it is skipped.  The defaults (480 s deadline, 75 s reserve): a hanging preflight pair and a hanging RCCL run
    still leave gloo and the exchange-free run their shares."""
    sys.path.insert(0, REPO)
    import bench
    assert bench.stage_limit(600, 480, 2, 75) == 480 - 150 - 10           # the rccl attempt cannot eat the fallbacks' time
    assert bench.stage_limit(90, 480, 3, 75) == 90                          # plenty left: the configured limit stands
    assert bench.stage_limit(600, 100, 1, 75) == 15 and bench.stage_limit(600, 89, 1, 75) == 0.0     # < 5 s: skipped
    left, spent = 480.0, []
    for conf, later in ((105, 3), (105, 2), (600, 1), (600, 0)):            # everything hangs: pf, pf, gloo, none
        lim = bench.stage_limit(conf, left, later, 75)
        assert lim > 0
        spent.append(lim); left -= lim + 10                                 # a stopped stage costs its limit + the stop margin
    assert left >= 0 and spent[-1] >= 75 - 10 - 1, (left, spent)


def test_a_worker_that_hangs_outside_the_collective_still_leaves_time_for_the_fallback():
    """a worker that hangs where no process-group timeout can reach it (GPU or library bring-up).  With ONE
    job deadline the RCCL attempt's limit is what the deadline leaves after the fallbacks' shares - not its own 600 s - so
    the gloo attempt still runs and its line is printed before the deadline; every stage records its limit."""
    r, lines, took = _run_stub(["--stub-hang-before-init", "rccl", "--job-deadline-s", "70", "--attempt-reserve-s", "14"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1 and took < 70, took
    c = json.loads(lines[0])["collective"]
    assert [(a["collective"], a["ok"]) for a in c["attempts"]] == [("rccl", False), ("gloo", True)]
    assert c["attempts"][0]["limit_s"] < 70 - 2 * 14 and "no result within" in c["attempts"][0]["reasons"][0]
    assert c["attempts"][1]["limit_s"] <= 70 - 14 and c["preflight"][0]["limit_s"] > 0
    assert c["budget"]["job_deadline_s"] == 70 and c["budget"]["used_s"] < 70 and c["used"] == "gloo"


def test_when_everything_hangs_the_job_ends_non_zero_at_the_deadline():
    """Preflights hang, then every worker hangs before it joins a process group: each stage is stopped at its share and the
    job returns a non-zero code within the deadline (plus the stop margin), with no line and no process left behind."""
    r, lines, took = _run_stub(["--stub-preflight", "hang", "--stub-hang-before-init", "all", "--job-deadline-s", "60",
                                "--attempt-reserve-s", "8", "--preflight-timeout-s", "6"])
    assert r.returncode != 0 and not lines, (r.returncode, lines)
    assert took < 60 + 15, took
    assert "within the job deadline of 60 s" in r.stderr and "gloo: " in r.stderr and "none: " in r.stderr
    import time
    left = []
    for _ in range(20):
        left = [ln for ln in subprocess.run(["pgrep", "-af", "stub-hang-before-init all"], capture_output=True, text=True).stdout.splitlines()
                if "pgrep" not in ln]
        if not left:
            break
        time.sleep(0.5)
    assert not left, f"processes left behind: {left}"


def test_when_every_fallback_fails_the_job_fails_within_the_limit():
    r, lines, took = _run_stub(["--stub-fail-collective", "all"])
    assert r.returncode != 0 and not lines
    assert "no fallback left" in r.stderr and took < 120, took


def test_an_explicit_collective_is_used_as_given():
    r, lines, _ = _run_stub(["--collective", "none"])
    assert r.returncode == 0, r.stderr[-3000:]
    c = json.loads(lines[0])["collective"]
    assert c["used"] == "none" and c["requested"] == "none" and c["preflight"] == [] and "gather" not in c and c["done_mask_exchange"] is None
    r, lines, _ = _run_stub(["--collective", "rccl", "--stub-fail-collective", "rccl"])
    assert r.returncode != 0 and not lines, "an explicit backend has no fallback"


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


def test_the_drivers_own_launcher_in_the_drivers_shape_four_ranks():
    """What the round-end driver runs for N > 1, to the letter - `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps 20 --warmup 5` - with the stub
    step (no GPU here) at four ranks: RANK / WORLD_SIZE / MASTER_* come from the launcher, nothing is self-launched, rank 0
    prints exactly one line, the short shape's bucket (16 steps) and its flushed tail (9 rows) both travel."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "4", "--steps", "20", "--warmup", "5", "--stub-step",
                        "--drones-per-gpu", "4099"], capture_output=True, text=True, timeout=400, env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, f"rank 0 must print exactly one JSON line, got {lines}"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["steps"] == 20 and out["warmup"] == 5 and out["gather_ok"] is True
    c = out["collective"]
    assert c["world_seen"] == 4 and c["rank_ids_gathered"] == [0, 1, 2, 3] and len(c["per_rank_ms_per_step"]["all"]) == 4
    assert c["gather"]["block_steps"] == 16 and c["gather"]["collectives_launched"] == 2
