#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the fused HIP stepper at 2^20 drones per GPU, dt = 1 ms.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus 8 ...                      # spawns its own 8 ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          # or under torchrun

Workload = BASELINE.json configs[2]: 1 048 576 drones per GPU, fp32, EMA-smoothed Gaussian stick
noise (the profile of /root/reference/tests/noise_smooth_test.py:6-12) generated on the device
BEFORE the timed region into a ring of action batches, so every timed step reads a different
16.8 MB action batch from HBM.  A "step" is one pass of the hot path over the batch: with
`--api step` (the headline) one kernel launch through the C ABI per step (fpv_step: what a closed
policy loop pays); with `--api rollout` the k-step kernel (fpv_step_n) advances a whole ring span per
launch with the drones held in registers (open-loop sticks: 16 + 117/k bytes per env-step - the action row every
step, the 112-byte state round trip and the 5 bytes of reward/done once per launch).
With N > 1 GPUs the drones are sharded contiguously (weak scaling: 2^20 per GPU) and each step's
bit-packed done mask is all-gathered over RCCL, asynchronously, bucketed and double-buffered.

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` and `cpu_baseline`.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

T_PROCESS_START = time.monotonic()       # the job deadline of an N-rank run counts from here (supervise_rank)
REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HOST_COST_LAUNCHES = 128  # host enqueue time is taken over at most this many launches: before the hardware queue can fill
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
VALU_PEAK_GINST = 256 * 4 * 32 * 2.4   # lane-instructions/ns: 256 CUs x 4 SIMD-32 x 2.4 GHz (same guide)
KERNEL_SOURCES = [os.path.join(REPO, "fpyv_amd", "csrc", f) for f in ("fpv_hip.hip", "fpv_math.h", "fpv_addr.h", "fpv_derive.h", "fpv_normal_table.h")]


def _strip_comments(src):
    """C/C++ source without comments and with runs of white space collapsed (string and character literals are kept
    verbatim), so that editing a comment does not make a measurement look stale - any change to code does."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c in "\"'":                                       # literal: copy to the closing quote, honouring escapes
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1]); i = j + 1
        elif src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c); i += 1
    return " ".join("".join(out).split())


def kernel_source_hash():
    """sha256 (16 hex digits) over the CODE of the kernel sources - comments and white space do not count - and the
    compiler flags the library is built with (a code-generation flag changes the kernels as surely as an edit does)."""
    h = hashlib.sha256()
    for p in KERNEL_SOURCES:
        with open(p, "r", encoding="utf-8") as f:
            h.update(_strip_comments(f.read()).encode("utf-8"))
        h.update(b"\0")
    from __graft_entry__ import HIPCC_FLAGS
    h.update(" ".join(f for f in HIPCC_FLAGS if not f.startswith("-W")).encode("utf-8"))
    return h.hexdigest()[:16]


def fit_launch_time(points):
    """Least squares of avg_launch_us(n) = floor_us + algorithmic bytes / streaming rate through `points` =
    [(drones, bytes_per_launch, avg_launch_us), ...] (three populations inside the Infinity Cache).  The fit is only as good
    as its points: `valid` is False - with the reason - when a point does not sit on the line (max residual > 0.5 us: a leg
    measured at idle clocks or bound by the host's launch loop), when the slope says the kernel streams faster than the HBM
    peak, or when the floor is not positive.  A reader takes floor_us / streaming_GBs from a valid fit only."""
    import numpy as np
    A = np.array([[1.0, b / 1e6] for _, b, _ in points])
    y = np.array([t for _, _, t in points], dtype=np.float64)
    (t0, slope), *_ = np.linalg.lstsq(A, y, rcond=None)
    resid = float(np.abs(A @ np.array([t0, slope]) - y).max())
    rate = float(1e3 / slope) if slope > 0 else float("inf")
    why = []
    if not np.isfinite(y).all() or (y <= 0).any():
        why.append("a point has no positive finite time")
    if resid > 0.5:
        why.append(f"max residual {resid:.2f} us > 0.5 us: the points are not on one line (a leg at idle clocks, or bound by the host's launch loop - compare host_enqueue_us with avg_launch_us)")
    if not slope > 0 or rate > HBM_PEAK_GBS:
        why.append(f"slope gives a streaming rate of {rate:.0f} GB/s, above the {HBM_PEAK_GBS:.0f} GB/s HBM peak")
    if not t0 > 0:
        why.append(f"floor {t0:.2f} us is not positive")
    return {"model": "avg_launch_us(n) = floor_us + algorithmic bytes / streaming rate (least squares; populations inside the Infinity Cache)",
            "drones": [int(n_) for n_, _, _ in points], "avg_launch_us": [float(t) for t in y],
            "floor_us": float(t0), "streaming_GBs": rate, "streaming_frac_of_peak": rate / HBM_PEAK_GBS,
            "max_residual_us": resid, "valid": not why, "invalid_reason": "; ".join(why) or None}


def hbm_roofline(value_per_gpu, bytes_per_env_step, drones, steps_per_launch, avg_launch_s):
    """The two clocks of one line, named.  `achieved` / `frac` FOLLOW `value`: algorithmic bytes per env-step x env-steps/s of one
    GPU by the wall clock of the whole timed region (barrier to synchronise, the tail after the last launch included) - a reader
    recomputes them from `value` alone: frac = value / n_gpus x algorithmic_bytes_per_env_step / 1e9 / peak.  `achieved_events` /
    `frac_events` are the same bytes over the HIP-event time from the first launch's start to the last launch's end
    (avg_launch_us): the kernel chain without the host's synchronise tail, which is what the rocprofv3 kernel trace sees.  In a
    2000-step run the two agree within a fraction of a per cent; in the driver's 20-step shape the ~20 us tail is 5 % of the region."""
    achieved = bytes_per_env_step * value_per_gpu / 1e9
    achieved_events = bytes_per_env_step * drones * steps_per_launch / avg_launch_s / 1e9
    return {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "achieved_events": achieved_events, "frac_events": achieved_events / HBM_PEAK_GBS}


def _lib_step_grid(n):
    """workgroups of one single-step launch: n drones in blocks of 128, whole rounds of the eight XCDs (csrc/fpv_hip.hip step_grid)"""
    return (n + 8 * 128 - 1) // (8 * 128) * 8


def xcd_map_probe(dev, blocks, launches=16):
    """Which XCD runs which workgroup (fpv_diag_xcd_map: the XCC_ID register), over `launches` launches of the step kernel's grid:
    the rotation's L2 tier assumes xcd(b) = (b + s) mod 8 with the same s for every launch of a chain.  HIP promises neither; this
    is the run-time look at it (a different answer costs cache reuse, never a result)."""
    import torch
    from fpyv_amd import _lib
    L = _lib.lib()
    out = torch.full((launches, blocks), 99, dtype=torch.int32, device=dev)
    for t in range(launches):
        _lib.check(L.fpv_diag_xcd_map(out[t].data_ptr(), blocks, torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize()
    m = out.cpu()
    b = torch.arange(blocks, dtype=torch.int32)
    shifts = [int(r[0]) % 8 for r in m]
    return {"blocks": blocks, "launches": launches, "round_robin_exact": all(bool(torch.equal(r, (b + s) % 8)) for r, s in zip(m, shifts)),
            "shift_stable": len(set(shifts)) == 1, "shift": shifts[0], "xcds_seen": len(set(m.flatten().tolist())),
            "note": "workgroup b of every launch ran on XCD (b + shift) mod 8: a drone block of the rotated traversal meets its own L2 again"}


def median(xs):
    xs = sorted(xs)
    m = len(xs) // 2
    return xs[m] if len(xs) % 2 else 0.5 * (xs[m - 1] + xs[m])


def library_hash():
    """sha256 (16 hex digits) of the built library, fpyv_amd/libfpv_hip.so: the kernels a counter pass ran ARE these bytes.  hipcc is
    deterministic (a forced rebuild of the same sources gives the same file), and code that is compiled out - an experiment hook
    behind an `#if` that is off - does not change it, so a measurement stays valid for exactly as long as the machine code does."""
    h = hashlib.sha256()
    with open(os.path.join(REPO, "fpyv_amd", "libfpv_hip.so"), "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()[:16]


def measurement_is_current(j):
    """A committed counter file (profiles/pmc_traffic.json, pmc_valu.json) belongs to the kernels of this tree when the library
    it was measured on is byte for byte the one that is built now, or when the kernel sources' code (comments stripped, with
    the compiler flags) still hashes to the value recorded with it."""
    # (hipcc derives its compilation-unit id from the output path, so the same sources built elsewhere give other bytes: the
    # source hash stays as the second key)
    if j.get("library_sha256_16") and j["library_sha256_16"] == library_hash():
        return True
    return j.get("kernel_source_sha256_16") == kernel_source_hash()


def usable_cpus():
    """CPUs this process can really run on: min(affinity mask, cgroup v2/v1 CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                     # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(params, seconds_budget=12.0):
    """The float64 C oracle (a port of Drone.step; the reference itself is Python and cannot
    travel) timed on this host: all cores via OpenMP, each thread walking its own contiguous tile of
    drones time-outer (state stays in L1/L2), built -O3 -march=native for THIS host; on a bounded
    sample of the same workload."""
    import numpy as np
    from oracle import oracle
    L = oracle.lib(native=True)
    # threads = the CPUs this process may actually use: the GPU box hands a job a cgroup share of the host
    # (about 16 CPUs per GPU), far fewer than os.cpu_count(); oversubscribing it only adds context switches
    usable = usable_cpus()
    threads = max(1, min(oracle.max_threads(), usable))
    n, chunk = 1 << 18, 32
    x = np.random.default_rng(1234).standard_normal((chunk, n, 4))
    acts = np.empty_like(x)
    s = np.zeros((n, 4))
    for t in range(chunk):                      # noise_smooth_test.py:11 recurrence
        s = 0.9 * s + 0.1 * x[t]
        acts[t] = s
    del x
    st = oracle.drone_initial_state(n, params.init_position, params.init_velocity, params.init_orientation_deg)
    oracle.drone_run(params, st, acts[:2], threads=threads, native=True)          # warm-up (thread pool, page faults)
    steps_done, t0 = 0, time.perf_counter()
    while True:
        oracle.drone_run(params, st, acts, threads=threads, native=True)
        steps_done += acts.shape[0]
        el = time.perf_counter() - t0
        if el > seconds_budget:
            break
    all_cores = n * steps_done / el
    n1 = n // 16
    st1 = oracle.drone_initial_state(n1, params.init_position, params.init_velocity, params.init_orientation_deg)
    a1 = np.ascontiguousarray(acts[:, :n1])
    oracle.drone_run(params, st1, a1[:2], threads=1, native=True)
    t0 = time.perf_counter()
    reps = 4
    for _ in range(reps):
        oracle.drone_run(params, st1, a1, threads=1, native=True)
    one_core = n1 * a1.shape[0] * reps / (time.perf_counter() - t0)
    del L
    # the host's best: the same float64 arithmetic with the DRONES as the vector axis (oracle/fpv_oracle_simd.c: SoA tiles of
    # 256 drones, `omp simd` over the tile, libmvec sin / cos, -O3 -march=native), all usable cores, time-outer per tile;
    # checked to 1e-12 against the scalar oracle in tests/test_numpy_port.py.  ~5 s
    simd = None
    try:
        sts = oracle.drone_initial_state(n, params.init_position, params.init_velocity, params.init_orientation_deg)
        oracle.drone_run_simd(params, sts, acts[:2], threads=threads, native=True)
        s_steps, t0 = 0, time.perf_counter()
        while True:
            oracle.drone_run_simd(params, sts, acts, threads=threads, native=True)
            s_steps += acts.shape[0]
            s_el = time.perf_counter() - t0
            if s_el > 5.0:
                break
        simd_all = n * s_steps / s_el
        sts1 = oracle.drone_initial_state(n1, params.init_position, params.init_velocity, params.init_orientation_deg)
        oracle.drone_run_simd(params, sts1, a1[:2], threads=1, native=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            oracle.drone_run_simd(params, sts1, a1, threads=1, native=True)
        simd_one = n1 * a1.shape[0] * reps / (time.perf_counter() - t0)
        isa = "unknown"
        try:
            fl = open("/proc/cpuinfo").read()
            isa = "AVX-512" if " avx512f" in fl else "AVX2" if " avx2" in fl else "AVX" if " avx " in fl else "SSE2"
        except OSError:
            pass
        simd = {"value": simd_all, "unit": "env-steps/s", "kind": "port", "cores": threads, "one_thread_value": simd_one, "vector_isa_of_host": isa,
                "over_scalar_port": simd_all / all_cores,
                "sample": f"{n} drones x {s_steps} steps of the same sticks, oracle/fpv_oracle_simd.c: float64, structure-of-arrays tiles of 256 drones, "
                          f"`#pragma omp simd` over the drones of a tile (gcc -O3 -march=native -ffp-contract=off, libmvec sin / cos), OpenMP over tiles with "
                          f"{threads} threads, time-outer inside a tile; same arithmetic as the scalar port (1e-12, tests/test_numpy_port.py)"
                          + ("" if simd_all >= all_cores else "; SLOWER than the scalar port on this host - kept as evidence")}
    except Exception as e:                      # noqa: BLE001  (a missing libmvec / compiler on some host must not cost the line)
        simd = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
    # the "idiomatic Python" baseline - the same step written with NumPy over a drone axis
    # (oracle/numpy_port.py, checked against the C oracle in tests/test_numpy_port.py); ~3 s of it
    from oracle import numpy_port
    nn = 1 << 16
    Sn = numpy_port.initial_state(nn, params.init_position, params.init_velocity, params.init_orientation_deg)
    an = np.ascontiguousarray(acts[:, :nn])
    numpy_port.step(params, Sn, an[0])
    t0, np_steps = time.perf_counter(), 0
    while time.perf_counter() - t0 < 3.0:
        numpy_port.step(params, Sn, an[np_steps % an.shape[0]])
        np_steps += 1
    numpy_rate = nn * np_steps / (time.perf_counter() - t0)
    scalar = {"value": all_cores, "unit": "env-steps/s", "cores": threads, "kind": "port", "one_thread_value": one_core,
              "sample": f"{n} drones x {steps_done} steps of EMA-noise sticks, float64 C restatement of Drone.step "
                        f"(oracle/fpv_oracle.c, gcc -O3 -march=native -fno-tree-vectorize: one drone at a time), OpenMP: {threads} threads each walking a "
                        f"contiguous drone tile time-outer; 1-thread rate {one_core:.3e} env-steps/s (speed-up {all_cores / one_core:.1f}x)"}
    best_is_simd = bool(simd.get("value")) and simd["value"] > all_cores
    best = simd if best_is_simd else scalar
    # the reported baseline is the host's BEST leg (a scalar per-drone port flatters the GPU); both legs and the
    # NumPy one stay on the line
    return {"value": best["value"], "unit": "env-steps/s", "cores": threads, "kind": "port",
            "leg": "simd_across_drones" if best_is_simd else "scalar_per_drone",
            "sample": best["sample"] + f"; {usable} usable of {os.cpu_count()} logical CPUs (nproc / cgroup quota); reference's own Python Drone.step, timed in the "
                      f"build container only (it cannot travel): 2.7e3-4.0e3 env-steps/s on one core (profiles/archive/r01_reference_python_timing.json)",
            "one_thread_value": best["one_thread_value"], "host_cpus": os.cpu_count(), "usable_cpus": usable, "threads_used": threads,
            "scalar_per_drone": scalar,
            "simd_across_drones": simd,
            "numpy_vectorised": {"value": numpy_rate, "unit": "env-steps/s", "kind": "port",
                                 "sample": f"{nn} drones x {np_steps} steps, oracle/numpy_port.py (float64 NumPy over a drone axis: the idiomatic-Python "
                                           f"batching of Drone.step, SURVEY 8d iii), NumPy {np.__version__} as it threads itself on this host"},
            "scaling_limiter": "the job's CPU share (cgroup quota / affinity), not the code: each thread owns a contiguous drone "
                               "tile whose state stays in its L1/L2, and there is no shared write"}


def spawn_ranks(n_ranks, argv, port=None, python=sys.executable, wall_limit_s=1500.0):
    """`bench.py --gpus N` outside torchrun: be the launcher - start N fresh child processes (one per rank, each of
    which becomes that rank's GPU-free supervisor, see supervise_rank) BEFORE this process touches the GPU, forward
    rank 0's stdout (the one JSON line), return the first non-zero exit code.
    Never re-execs: the parent only watches.  ALL children are polled together: the first rank that exits non-zero
    (no GPU, an import error, the non-finite-state assert) ends the job at once - its siblings would otherwise sit in
    the rendezvous, an all-gather or a barrier until the backend's own 10-30 min timeout - and `wall_limit_s` bounds
    the whole run (exit code 124, like timeout(1))."""
    import socket
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # a caller's value wins (it is recorded in collective.ipc_mode)
        procs.append(subprocess.Popen([python, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=(None if r == 0 else subprocess.DEVNULL)))

    def stop(ps):
        for q in ps:                       # exactly the PIDs started above: terminate, then kill what ignores it
            if q.poll() is None:
                q.terminate()
        t_end = time.monotonic() + 10.0
        for q in ps:
            try:
                q.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    deadline = time.monotonic() + wall_limit_s
    live = dict(enumerate(procs))
    while live:
        for r, q in list(live.items()):
            code = q.poll()
            if code is None:
                continue
            del live[r]
            if code != 0:
                print(f"bench.py: rank {r} exited with {code}; stopping the other {len(live)} rank(s)", file=sys.stderr)
                stop(list(live.values()))
                return abs(code) or 1
        if live and time.monotonic() > deadline:
            print(f"bench.py: ranks {sorted(live)} still running after {wall_limit_s:.0f} s; stopping them", file=sys.stderr)
            stop(list(live.values()))
            return 124
        if live:
            time.sleep(0.05)
    return 0


def stage_limit(configured_s, time_left_s, later_attempts, reserve_s, margin_s=10.0):
    """Wall limit of the next stage (a preflight or an attempt) of an N-rank job under ONE deadline: the configured limit,
    cut to the time left minus what the stages after it must keep - `reserve_s` per attempt still in the plan, so that the
    last resort (gloo, then no exchange at all) can still start, run and print - and minus `margin_s` for stopping a child
    that hangs (terminate, grace, kill).  Returns 0.0 when less than 5 s would remain: the stage is skipped."""
    lim = min(float(configured_s), float(time_left_s) - later_attempts * float(reserve_s) - margin_s)
    return lim if lim >= 5.0 else 0.0


def supervise_rank(args, raw_argv):
    """One rank of an N-rank job as the launcher started it - by torchrun (the driver's way) or by spawn_ranks.  This
    process stays GPU-FREE: it agrees with its peers over gloo on CPU tensors (fpyv_amd.dist.RankSupervisor) and runs
    everything that touches the GPU in fresh child processes it can stop by their PID, so the job cannot come back
    empty-handed because RCCL would not start on this node - or because something hangs:
      1. RCCL preflight in fresh children (init + one all-gather of the rank ids, wall limit) with the caller's
         HSA_ENABLE_IPC_MODE_LEGACY; when that fails, once more with the other value;
      2. the worker (`--worker --collective rccl`) per rank; if the preflight failed, or the workers fail or hang, fresh
         workers once more with the done mask over gloo (host-staged), and as the last resort with no exchange;
      3. rank 0 prints the accepted worker's JSON line, with what happened in collective.{requested, backend, ipc_mode,
         fallback_reason, preflight, attempts, budget}.  Non-zero exit only when every attempt failed.
    ONE deadline bounds the sum (`--job-deadline-s`, counted from this process's start): every stage's limit is
    min(its configured limit, time left - `--attempt-reserve-s` per attempt still in the plan - a stop margin), agreed
    across the ranks (the minimum), recorded as `limit_s`; a stage with less than 5 s left is skipped, and past the
    deadline the job ends with a non-zero code.  Children are always fresh processes - never an exec of one that has
    touched the GPU.  Loop being sharded: /root/reference/src/core/simulator.py:83-156."""
    from fpyv_amd import dist as fd
    watch_parent()
    deadline = T_PROCESS_START + args.job_deadline_s
    with stdout_to_stderr():                     # gloo announces itself on stdout
        sup = fd.RankSupervisor(timeout_s=args.job_deadline_s + 120.0)
    rank = sup.rank
    ipc_caller = os.environ.get(fd.IPC_ENV)
    ipc_mode = ipc_caller
    stub_pf = args.stub_preflight if args.stub_step else None
    preflights, attempts, final = [], [], None
    reserve = args.attempt_reserve_s

    def limit_for(configured, later_attempts):
        # every rank's own clock, then the minimum over the ranks: all of them stop their children together
        return sup.agree_min(stage_limit(configured, deadline - time.monotonic(), later_attempts, reserve))

    def preflight(mode, later_attempts):
        conf = args.preflight_timeout_s
        allow = min(15.0, max(2.0, conf))                  # start-up of the child on top of the bring-up limit it is given
        lim = limit_for(conf + allow, later_attempts)      # the child's wall limit
        if lim <= 0:
            return {"ok": False, "ipc_mode": mode, "seconds": 0.0, "limit_s": 0.0, "reasons": ["skipped: no time left under the job deadline"], "library_version": None}
        inner = conf if lim >= conf + allow else max(2.0, lim - min(allow, 0.3 * lim))
        res = sup.preflight(mode, inner, stub_pf, wall_s=lim)
        res["limit_s"] = round(lim, 1)
        return res

    if args.rehearse_on_one_gpu:
        plan = ["gloo"]                          # RCCL refuses two ranks on one device
    elif args.collective != "auto":
        plan = [args.collective]
    else:
        plan = ["rccl", "gloo", "none"]
        if not args.no_preflight:
            res = preflight(ipc_caller, later_attempts=3)          # (a second preflight comes out of the rccl attempt's share)
            preflights.append(res)
            if not res["ok"]:
                alt = fd.other_ipc_mode(ipc_caller)
                if rank == 0:
                    print(f"bench.py: RCCL preflight failed with {fd.IPC_ENV}={ipc_caller!r} ({'; '.join(res['reasons'])[:300]}); trying {alt!r}", file=sys.stderr)
                res = preflight(alt, later_attempts=2)              # RCCL will only be tried if this one passes: keep gloo's and none's shares
                preflights.append(res)
                if res["ok"]:
                    ipc_mode = alt
                else:
                    plan = ["gloo", "none"]
    requested = "gloo" if args.rehearse_on_one_gpu else ("rccl" if args.collective == "auto" else args.collective)
    for k, mode in enumerate(plan):
        lim = limit_for(args.attempt_timeout_s, later_attempts=len(plan) - 1 - k)
        if lim <= 0:
            attempts.append({"collective": mode, "ipc_mode": ipc_mode, "ok": False, "seconds": 0.0, "limit_s": 0.0,
                             "reasons": ["skipped: no time left under the job deadline"]})
            continue
        port = sup.pick_port()
        cmd = [sys.executable, os.path.abspath(__file__)] + list(raw_argv) + ["--worker", "--collective", mode]
        r = fd.run_child(cmd, sup.child_env(port, ipc_mode), lim, capture_stdout=(rank == 0), on_start=sup.track)
        line, ok_local = None, r["rc"] == 0
        if rank == 0 and ok_local:
            cand = [ln for ln in r["stdout"].splitlines() if ln.lstrip().startswith("{")]
            try:
                line = json.loads(cand[-1])
            except (IndexError, ValueError):
                ok_local = False
        why = None if ok_local else (f"no result within {lim:.0f} s" if r["timed_out"] else f"exit code {r['rc']}: {r['stderr_tail']}")
        ok = sup.all_ok(ok_local)
        reasons = sup.gather_reasons(why)
        attempts.append({"collective": mode, "ipc_mode": ipc_mode, "ok": ok, "seconds": round(r["seconds"], 2), "limit_s": round(lim, 1), "reasons": reasons})
        if ok:
            final = line
            break
        if rank == 0:
            print(f"bench.py: the run with --collective {mode} failed ({'; '.join(reasons)[:400]})"
                  + ("; starting fresh workers with the next fallback" if mode != plan[-1] else "; no fallback left"), file=sys.stderr)
    done = bool(attempts) and attempts[-1]["ok"]
    used_s = time.monotonic() - T_PROCESS_START
    if rank == 0 and not done:
        print(f"bench.py: no attempt of the {sup.world}-rank job succeeded within the job deadline of {args.job_deadline_s:.0f} s ({used_s:.0f} s used): "
              + "; ".join(f"{x['collective']}: {'; '.join(x['reasons'])[:200]}" for x in attempts), file=sys.stderr)
    if rank == 0 and done:
        c = final.setdefault("collective", {})
        used = attempts[-1]["collective"]
        why = None
        if used != requested:
            failed_pf = [x for x in preflights if not x["ok"]]
            failed_at = [x for x in attempts if not x["ok"]]
            why = ("RCCL preflight failed: " + "; ".join(failed_pf[0]["reasons"]) if failed_pf and not any(x["collective"] == "rccl" for x in attempts)
                   else "run failed: " + "; ".join(failed_at[0]["reasons"]) if failed_at else "unknown")[:600]
        c.update(requested=requested, used=used, ipc_mode=ipc_mode, ipc_mode_of_caller=ipc_caller,
                 ipc_mode_source=os.environ.get("FPV_BENCH_IPC_MODE_SOURCE"), fallback_reason=why,
                 preflight=preflights, attempts=attempts,
                 budget={"job_deadline_s": args.job_deadline_s, "used_s": round(used_s, 1), "reserve_per_later_attempt_s": reserve,
                         "rule": "limit of a stage = min(configured, time left - reserve x attempts still in the plan - 10 s), minimum over ranks"})
        print(json.dumps(final), flush=True)
    sup.close()
    return 0 if done else 1


class stdout_to_stderr:
    """The driver reads exactly ONE JSON line on stdout; RCCL and gloo print banners there when the first
    communicator comes up, so fd 1 points at stderr while that happens."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def watch_parent():
    """A worker whose supervisor is gone (killed by the launcher's own limit) must not live on as an orphan holding a
    GPU: a daemon thread leaves as soon as the parent PID changes."""
    import threading
    ppid = os.getppid()

    def loop():
        while os.getppid() == ppid:
            time.sleep(0.5)
        os._exit(3)
    threading.Thread(target=loop, daemon=True).start()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--default-stream", action="store_true",
                    help="launch on the legacy default (null) stream instead of a stream of the run's own")
    ap.add_argument("--preheat-s", type=float, default=0.5,
                    help="seconds of untimed launches on a SCRATCH batch before the W warm-up steps, so the "
                         "GPU has left its idle clocks (the timed region is only K x ~25 us)")
    ap.add_argument("--aux-preheat-s", type=float, default=0.25,
                    help="the same time-based preheat before EACH auxiliary leg (sustained, beyond_mall, launch_time_fit), on that "
                         "leg's own buffers: each leg starts right after gigabytes were freed and allocated")
    ap.add_argument("--drones-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--ring", type=int, default=32, help="distinct pre-generated action batches")
    ap.add_argument("--api", choices=["rollout", "step", "rollout-launches"], default="step",
                    help="step (headline): one Python env.step() = one launch per step; rollout: the k-step kernel "
                         "(fpv_step_n), one launch per ring span; rollout-launches: k single-step launches from one C call")
    ap.add_argument("--partitions", type=int, default=1,
                    help="--api step only, N = 1: step the population as this many column partitions, each an independent kernel "
                         "chain on its own stream (FpvVecEnv.step_async / step_wait, the split-phase API a closed-loop caller can "
                         "use): the chains hide part of each other's per-launch floor.  A labelled line, not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-beyond-mall", action="store_true",
                    help="skip the extra short run at 2^23 drones (state 470 MB > the 256 MiB Infinity Cache)")
    ap.add_argument("--ceiling", type=float, default=100.0, help="auto-reset when |z| exceeds this (m)")
    ap.add_argument("--no-auto-reset", action="store_true")
    ap.add_argument("--fp16-state", action="store_true",
                    help="BASELINE configs[3]: v,q,rates,thrust stored as binary16 (89 B/env-step); not the headline")
    ap.add_argument("--racer", choices=["written", "omega_dt"], default=None,
                    help="time the Racer.step kernel instead (mode B; not the headline)")
    ap.add_argument("--no-gather", action="store_true", help="skip the done-mask all-gather (N > 1)")
    ap.add_argument("--gather-block", type=int, default=0,
                    help="steps of done masks bucketed into one all-gather (N > 1); 0 = sized to the run: every collective "
                         "costs the launch stream ~25 us whatever its size (measured, profiles/archive/r02_exp_gather_block.log), so long "
                         "runs use 64-step buckets (8 MiB per rank, 1.5 %% of the time) and runs under 256 steps 16-step buckets, "
                         "whose unfilled tail - gathered inside the timed region - stays small")
    ap.add_argument("--gather-returns", action="store_true",
                    help="N > 1: also all-gather the per-drone last episode return (fp32) once per bucket; turns on "
                         "episode bookkeeping (+16 B per env-step), so it is off for the headline numbers")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the RCCL process group + all-gather path even with one rank")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="TEST ONLY (tests/test_gpu_boundary.py): run the real N-rank path - shards, per-rank stick streams, kernels, "
                         "bucketed done-mask all-gather, flush, MAX over ranks - with every rank on GPU 0 and the gloo backend (RCCL "
                         "refuses two ranks on one device); the line is marked data=rehearsal and its value means nothing")
    ap.add_argument("--dump-gathered", default=None,
                    help="with --rehearse-on-one-gpu: rank 0 saves the gathered done masks of the last bucket and every rank its "
                         "own final state to this directory (the test compares them with a single-process run)")
    ap.add_argument("--spawn-timeout-s", type=float, default=0.0,
                    help="self-launched ranks (--gpus N outside a launcher): wall-clock limit of the whole job; 0 (default) = the job "
                         "deadline + 60 s (the supervisors end the job themselves at the deadline; this is the launcher's backstop)")
    ap.add_argument("--collective", choices=["auto", "rccl", "gloo", "none"], default="auto",
                    help="N > 1: what carries the done mask and the job's barriers.  auto (default): RCCL if a preflight in fresh "
                         "child processes brings it up (with the caller's HSA_ENABLE_IPC_MODE_LEGACY, then once with the other value), "
                         "else gloo (host-staged), else no exchange at all - the line says which (collective.backend, "
                         "collective.fallback_reason).  An explicit value is used as given, without preflight or fallback")
    ap.add_argument("--worker", action="store_true",
                    help="INTERNAL: this process is the GPU-touching worker of one rank, started by its supervisor (supervise_rank)")
    ap.add_argument("--no-preflight", action="store_true", help="N > 1, --collective auto: skip the RCCL preflight (fallbacks stay)")
    ap.add_argument("--preflight-timeout-s", type=float, default=90.0,
                    help="wall limit of one RCCL preflight (init + one all-gather in fresh children; torch is already in the page "
                         "cache by then, so this is GPU + RCCL bring-up time only)")
    ap.add_argument("--attempt-timeout-s", type=float, default=600.0, help="wall limit of one attempt of the N-rank run (cut to what the job deadline leaves)")
    ap.add_argument("--job-deadline-s", type=float, default=480.0,
                    help="N > 1: ONE wall-clock deadline for the whole supervised job, counted from the start of the rank's process: "
                         "every preflight and attempt gets min(its own limit, the time left minus --attempt-reserve-s per attempt still "
                         "in the plan), so that a hang anywhere still leaves the last fallback time to print its line")
    ap.add_argument("--attempt-reserve-s", type=float, default=75.0,
                    help="N > 1: what every attempt still in the plan keeps for itself (process start, import, rendezvous, the short run)")
    ap.add_argument("--pg-timeout-s", type=float, default=180.0, help="process-group timeout inside a worker: a rank whose peer died leaves its collective after this long")
    ap.add_argument("--sustained-steps", type=int, default=2000,
                    help="N = 1, --api step: after the contract's K timed steps, this many more launches at the headline size, HIP-event "
                         "timed (roofline.sustained): the steady state next to the driver's short shape; 0 = skip")
    ap.add_argument("--stub-preflight", choices=["ok", "fail", "hang"], default="ok",
                    help="TEST ONLY (with --stub-step): what the preflight children do")
    ap.add_argument("--stub-fail-collective", choices=["", "rccl", "gloo", "all"], default="",
                    help="TEST ONLY (with --stub-step): workers started with this --collective (or all) exit 1 after the rendezvous")
    ap.add_argument("--stub-hang-collective", choices=["", "rccl", "gloo"], default="",
                    help="TEST ONLY (with --stub-step): workers started with this --collective never come back (a collective that hangs)")
    ap.add_argument("--stub-hang-before-init", choices=["", "rccl", "gloo", "all"], default="",
                    help="TEST ONLY (with --stub-step): workers started with this --collective (or all) hang OUTSIDE any collective - before "
                         "they join the process group, the way a GPU or library bring-up that never returns would")
    ap.add_argument("--stub-fail-rank", type=int, default=-1,
                    help="TEST ONLY (tests/test_bench_spawn.py): this rank exits 1 before it joins the process group")
    ap.add_argument("--stub-step", action="store_true",
                    help="TEST ONLY (tests/test_bench_spawn.py): no GPU, gloo, a no-op step - exercises the spawn, rendezvous, "
                         "barrier, max-over-ranks and JSON plumbing; the line it prints is marked data=stub and measures nothing")
    return ap.parse_args(argv)


def collective_report(dist, world_env, rank, dev, gather, local_ms_per_step, collective=None, physics_ms=None):
    """What lets a reader of the N-rank line verify that the collective really spanned N ranks without trusting the
    headline number: the world size the process group reports once it is up, the backend and its library version, an
    all-gather of every rank's id (must come back as 0..N-1, in order) and of every rank's own ms per step."""
    import torch
    out = {"backend": None, "world_env": world_env, "world_seen": 1, "rank_ids_gathered": [0], "library_version": None,
           "ipc_mode": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
           "done_mask_exchange": None if gather is None else {"rccl": "RCCL all-gather over xGMI", "gloo": "gloo all-gather, host-staged (fallback)"}.get(collective, collective)}
    if dist is None or not dist.is_initialized():
        return out
    world = dist.get_world_size()
    out["backend"], out["world_seen"] = str(dist.get_backend()), world
    ids = torch.full((1,), rank, dtype=torch.int64, device=dev)
    got = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(got, ids)
    ms = torch.tensor([local_ms_per_step], dtype=torch.float64, device=dev)
    all_ms = torch.empty(world, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(all_ms, ms)
    out["rank_ids_gathered"] = [int(x) for x in got.cpu()]
    per = [float(x) for x in all_ms.cpu()]
    out["per_rank_ms_per_step"] = {"min": min(per), "max": max(per), "all": per}
    if physics_ms is not None:
        # the same kernels with the mask exchange switched off, HIP-event timed on each rank's own launch stream after
        # the timed region: what the physics alone costs per step on every GPU of the job, whatever carried the masks
        ms = torch.tensor([physics_ms], dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(all_ms, ms)
        per = [float(x) for x in all_ms.cpu()]
        out["per_rank_physics_only_ms_per_step"] = {"min": min(per), "max": max(per), "all": per}
    if out["backend"] == "nccl":
        try:
            out["library_version"] = "RCCL/NCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:                      # noqa: BLE001  (a missing version query must not cost the line)
            out["library_version"] = f"unavailable ({type(e).__name__})"
    else:
        out["library_version"] = f"torch {torch.__version__} {out['backend']}"
    if gather is not None:
        row_bytes = gather.local[0][0].numel() * gather.local[0].element_size()
        out["gather"] = {"block_steps": gather.block, "bytes_per_bucket": gather.block * row_bytes,
                         "bytes_per_step_per_rank": row_bytes, "collectives_launched": gather.launched,
                         "what": "bit-packed done mask, one 64-bit word per 64 drones, all_gather_into_tensor(async_op=True), double-buffered"}
    return out


def run_stub(args, world, rank, collective="rccl"):
    """The N-rank plumbing on CPU tensors over gloo with a no-op step (see --stub-step).  `collective` is what the
    supervisor asked for ("rccl" is played by gloo here: there is no GPU); --stub-fail-collective makes the workers of
    one kind fail after the rendezvous, the way a collective that does not come up would."""
    import torch
    import torch.distributed as dist
    from fpyv_amd.dist import DoneGather, gloo_env_fixups
    if world > 1:
        gloo_env_fixups()
        with stdout_to_stderr():                 # gloo, like RCCL, announces itself on stdout
            dist.init_process_group(backend="gloo")
            dist.barrier()
        if args.stub_fail_collective in (collective, "all"):
            print(f"stub: the {collective} collective of rank {rank} fails", file=sys.stderr)
            raise SystemExit(1)
        if args.stub_hang_collective == collective and rank == world - 1:
            print(f"stub: rank {rank} hangs in the {collective} collective", file=sys.stderr)
            time.sleep(3600)
    words = (args.drones_per_gpu + 63) // 64
    gather = DoneGather((words,), torch.int64, "cpu", block=args.gather_block) if world > 1 and collective != "none" and not args.no_gather else None
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for t in range(args.steps):
        if gather is not None:
            gather.row(t).fill_(rank * 1000 + t)
            gather.step_done(t)
    if gather is not None:
        gather.flush(args.steps - 1)
        gather.drain()
    local = time.perf_counter() - t0
    elapsed = local
    if world > 1:
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ok = True
        if gather is not None:
            last = gather.result((args.steps - 1) // args.gather_block)        # a flushed bucket holds only its filled rows
            ok = all(int(last[r, (args.steps - 1) % args.gather_block, 0]) == r * 1000 + args.steps - 1 for r in range(world))
    else:
        ok = True
    coll = collective_report(dist if world > 1 else None, world, rank, "cpu", gather, local * 1e3 / max(args.steps, 1), collective=collective)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": 0.0, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "data": "stub", "gather_ok": bool(ok), "ms_per_step": elapsed * 1e3 / max(args.steps, 1),
                          "collective": coll}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main(argv=None):
    # multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC): without it RCCL fails
    # with "hipIpcGetMemHandle: invalid argument".  Set before anything loads the HIP runtime.
    os.environ.setdefault("FPV_BENCH_IPC_MODE_SOURCE", "caller's environment" if "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ else "bench.py default")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # setdefault: a caller's value wins and is recorded (collective.ipc_mode)
    args = parse_args(argv)
    raw_argv = sys.argv[1:] if argv is None else list(argv)
    in_rank = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not in_rank:
        # self-launch: nothing in this process has touched the GPU yet (device_count() does not initialise it)
        if not args.stub_step and not args.rehearse_on_one_gpu:
            import torch
            have = torch.cuda.device_count()
            if have < args.gpus:
                raise SystemExit(f"bench.py --gpus {args.gpus} needs {args.gpus} GPUs on this node, found {have}; "
                                 f"run with --gpus {max(have, 1)} (or --force-dist to rehearse the collective path on one GPU)")
        raise SystemExit(spawn_ranks(args.gpus, raw_argv, wall_limit_s=args.spawn_timeout_s if args.spawn_timeout_s > 0 else args.job_deadline_s + 60.0))

    if args.gather_block <= 0:
        args.gather_block = 64 if args.steps + args.warmup >= 256 else 16
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if rank == args.stub_fail_rank:
        raise SystemExit(1)                                  # TEST ONLY: a rank that dies before the rendezvous
    if world > 1 and not args.worker:
        raise SystemExit(supervise_rank(args, raw_argv))     # this process stays GPU-free; the worker is a fresh child
    if args.worker:
        watch_parent()
    collective = "gloo" if args.rehearse_on_one_gpu else ("rccl" if args.collective == "auto" else args.collective)
    if args.stub_step:
        if args.worker and args.stub_hang_before_init in (collective, "all"):
            print(f"stub: the {collective} worker of rank {rank} hangs before it joins the process group", file=sys.stderr)
            time.sleep(3600)
        raise SystemExit(run_stub(args, world, rank, collective))

    import torch
    import torch.distributed as dist
    from fpyv_amd import load_params, sticks
    from fpyv_amd import _lib as _lib_mod
    from fpyv_amd.dist import DoneGather
    from fpyv_amd.env import DroneBatch, RacerBatch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the stepper has no CPU path")
    if args.rehearse_on_one_gpu:
        local_rank = 0                                   # every rank on GPU 0 (at most 6 processes may share the card)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank} needs GPU {local_rank}, this node has {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if not args.default_stream:
        # the whole run on a stream of its own instead of the legacy default stream: a launch on the null stream orders
        # itself against every blocking stream of the process (a collective library's internal ones, for instance)
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    multi = world > 1 or args.force_dist
    if multi:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
        import datetime
        from fpyv_amd.dist import gloo_env_fixups
        pg_timeout = datetime.timedelta(seconds=args.pg_timeout_s)
        with stdout_to_stderr():
            if collective == "rccl":
                dist.init_process_group(backend="nccl", device_id=dev, timeout=pg_timeout)
            else:                                    # "gloo": the masks travel host-staged; "none": barriers and the MAX only
                gloo_env_fixups()
                dist.init_process_group(backend="gloo", timeout=pg_timeout)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()

    n = args.drones_per_gpu
    # dt = 1 ms; lanes that hit the ground or leave |z| <= ceiling are re-initialised in-kernel
    # (BASELINE configs[4] semantics; costs no extra bytes), so a long run stays a flight workload
    params = load_params(fps=1000, ceiling=args.ceiling)

    def make_env(n_drones, with_bits):
        if args.racer:
            pid = [[0.004, 0.02, 1e-6], [0.003, 0.01, 2e-6], [0.002, 0.005, 0.0]]
            import numpy as np
            rp = params.replace(mode=1, racer_pid=np.asarray(pid), racer_omega_dt=(args.racer == "omega_dt"), ceiling=args.ceiling)
            return RacerBatch(rp, n_drones, device=dev, auto_reset=not args.no_auto_reset, with_done_bits=with_bits)
        return DroneBatch(params, n_drones, device=dev, auto_reset=not args.no_auto_reset, with_accel=False,
                          with_done_bits=with_bits, fp16_state=args.fp16_state,
                          track_episodes=bool(multi and args.gather_returns))

    venv = None
    if args.partitions > 1:
        if args.api != "step" or multi or args.racer or args.fp16_state:
            raise SystemExit("--partitions needs --api step on one GPU with fp32 drone state (no --force-dist / --racer / --fp16-state)")
        from fpyv_amd.env import FpvVecEnv
        venv = FpvVecEnv(params, num_envs=n, device=dev, auto_reset=not args.no_auto_reset, track_episodes=False,
                         partitions=args.partitions)
        venv.reset()
        env = venv.batch
    else:
        env = make_env(n, multi)
        env.reset()

    total = args.steps + args.warmup
    ring = max(1, min(args.ring, total))
    # seeded per rank so that every GPU of the job sees different sticks (global drone ids differ)
    actions = sticks.ema_noise_device(ring, n, dev, seed=1234 + rank)

    gather = None
    words = (n + 63) // 64
    if multi and not args.no_gather and collective != "none":
        gather = DoneGather((words,), torch.int64, dev, block=args.gather_block)
        gather.warm_up()                    # RCCL's first-use costs stay out of a short timed region
    returns_all, returns_work = None, None
    if gather is not None and args.gather_returns:
        returns_all = torch.zeros(dist.get_world_size() * n, dtype=torch.float32, device=dev)

    launches = [0]
    part_slices = {}
    ring_rows = {}
    part_ranges = [venv.partition_range(kk) for kk in range(venv.partitions)] if venv is not None else []

    def run_on(e, acts, k, t_base, g):
        """k steps.  api=step: k launches; api=rollout: one k-step-kernel launch per ring span (and per gather bucket)."""
        nonlocal returns_work
        rlen = acts.shape[0]
        if venv is not None and e is env:
            # split phase: every partition's chain gets its step t; nothing joins the chains between steps (a policy
            # would sit between step_wait(part) and step_async(part) of ONE partition while the other one steps)
            sl = part_slices.get(acts.data_ptr())
            if sl is None:                          # built once per ring (a setdefault would build the 2 x ring slices on every call)
                sl = part_slices[acts.data_ptr()] = [[acts[r][lo:hi] for lo, hi in part_ranges] for r in range(rlen)]
            for t in range(t_base, t_base + k):
                row = sl[t % rlen]
                for kk in range(venv.partitions):
                    venv.step_async(kk, row[kk], ready=True)
                launches[0] += 1                    # one full-population step (= `partitions` overlapping launches)
            for kk in range(venv.partitions):
                venv.step_wait(kk)                  # the caller's stream (and its timing events) follow every chain
            return
        if args.api == "step":
            rows = ring_rows.get(acts.data_ptr())
            if rows is None:                        # the ring's batches as tensors of their own, made once: what a policy hands over
                rows = ring_rows[acts.data_ptr()] = [acts[r] for r in range(rlen)]
            for t in range(t_base, t_base + k):
                if g is not None:
                    e.set_done_bits_target(g.row_ptr(t))
                e.step(rows[t % rlen], return_imu=False)
                launches[0] += 1
                if g is not None:
                    g.step_done(t)
                    if returns_all is not None and (t + 1) % args.gather_block == 0:
                        if returns_work is not None:
                            returns_work.wait()
                        returns_work = dist.all_gather_into_tensor(returns_all, e.last_return, async_op=True)
            return
        fused = args.api == "rollout"
        t = t_base
        while t < t_base + k:
            r0 = t % rlen
            span = min(rlen - r0, t_base + k - t)
            if g is not None:                       # a span never crosses a gather bucket: the bucket rows are the mask rows
                span = min(span, g.block - t % g.block)
                e.set_done_bits_target(g.row_ptr(t), stride_words=words)
            e.rollout(acts[r0:r0 + span], fused=fused)
            launches[0] += 1 if fused else span
            t += span
            if g is not None and t % g.block == 0:
                g.step_done(t - 1)

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    if args.preheat_s > 0:
        scratch = make_env(n, False)
        scratch.reset()
        t_end = time.perf_counter() + args.preheat_s
        while time.perf_counter() < t_end:
            scratch.rollout(actions, fused=False)
            torch.cuda.synchronize()
        del scratch
    run_on(env, actions, args.warmup, 0, gather)
    # the two timing events exist (and have been recorded once) before the clock starts: an event is created lazily at
    # its first record, and the process's first timed event also sets up the runtime's profiling signals - tens of
    # microseconds that belong to the measuring apparatus, not to the 20 steps the driver's shape times
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    ev1.record()
    fence()
    ev0.elapsed_time(ev1)
    launches[0] = 0
    t0 = time.perf_counter()
    ev0.record()                      # torch's current stream == the stream the kernels are launched on
    # the host's cost per step, taken over the first steps only: in a long run the host runs ahead until the hardware queue is
    # full and then waits for the GPU - its time per step is then the GPU's, which says nothing about the host
    k_host = min(args.steps, HOST_COST_LAUNCHES) if (venv is None and args.api == "step") else args.steps     # (k-step launches are not cut)
    run_on(env, actions, k_host, args.warmup, gather)
    host_enqueue_s = time.perf_counter() - t0
    if k_host < args.steps:
        run_on(env, actions, args.steps - k_host, args.warmup + k_host, gather)
    ev1.record()
    if gather is not None:
        gather.flush(args.warmup + args.steps - 1)
        gather.drain()
        if returns_work is not None:
            returns_work.wait()
    # every rank stops its own clock when ITS work (kernels + its part of the collectives) is complete; the job's time
    # is the MAX over ranks (all-reduced below), and the closing barrier follows the clock instead of sitting inside it
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    dev_ms = ev0.elapsed_time(ev1)
    n_launches = launches[0]

    local_elapsed = elapsed
    coll = None
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    assert bool(torch.isfinite(env.state).all()), "non-finite state after the benchmark"
    if args.dump_gathered and gather is not None:
        import numpy as np
        os.makedirs(args.dump_gathered, exist_ok=True)
        last_t = args.warmup + args.steps - 1
        if rank == 0:        # [world, rows, words] masks of the bucket that holds the last step
            np.save(os.path.join(args.dump_gathered, "gathered_last_bucket.npy"), gather.result(last_t // args.gather_block).cpu().numpy())
        np.save(os.path.join(args.dump_gathered, f"state_rank{rank}.npy"), env.state.cpu().numpy())

    def timed_leg(e, acts, k, t_base, repeats=1, c_loop=False):
        """`repeats` x k more steps of the same kind on the launch stream, each repeat HIP-event timed, no exchange.
        c_loop: the launches come from one C call per ring span (fpv_rollout - the same single-step kernel, ~1 us of host
        time per launch) instead of one Python env.step() each, so that a slow host cannot bound the leg.
        Returns {"avg_launch_us": median over the repeats, "repeats_us", "launches" (per repeat), "host_enqueue_us":
        median host time per launch spent enqueueing (perf_counter around the launch loop, before the synchronise)}."""
        us_all, host_all, nl = [], [], 0
        for rep in range(repeats):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            before = launches[0]
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            a0.record()
            if c_loop:
                rlen, t = acts.shape[0], t_base + rep * k
                while t < t_base + (rep + 1) * k:
                    r0 = t % rlen
                    span = min(rlen - r0, t_base + (rep + 1) * k - t)
                    e.rollout(acts[r0:r0 + span], fused=False)
                    launches[0] += span
                    t += span
                h1, nh = time.perf_counter(), launches[0] - before
            else:
                kh = min(k, HOST_COST_LAUNCHES) if (args.api == "step" and not (venv is not None and e is env)) else k   # (see the timed region)
                run_on(e, acts, kh, t_base + rep * k, None)
                h1, nh = time.perf_counter(), launches[0] - before
                if kh < k:
                    run_on(e, acts, k - kh, t_base + rep * k + kh, None)
            a1.record()
            torch.cuda.synchronize()
            nl = launches[0] - before
            launches[0] = before
            us_all.append(a0.elapsed_time(a1) * 1e3 / max(nl, 1))
            host_all.append((h1 - h0) * 1e6 / max(nh, 1))
        us, host = median(us_all), median(host_all)
        return {"avg_launch_us": us, "repeats_us": us_all, "launches": nl, "host_enqueue_us": host,
                "host_bound": bool(host > 0.9 * us)}

    def leg_preheat(e, acts):
        """Time-based, like the headline's --preheat-s, on the leg's OWN buffers (a leg starts right after gigabytes were freed
        and allocated; 40 launches are 0.5 ms - not enough for the clocks): untimed C-loop launches for --aux-preheat-s."""
        if args.aux_preheat_s <= 0:
            return
        t_end = time.perf_counter() + args.aux_preheat_s
        while time.perf_counter() < t_end:
            e.rollout(acts, fused=False)
            torch.cuda.synchronize()

    physics_ms = None
    if multi:
        # the physics alone, per rank, after the timed region: the mask goes back to the batch's own word row (a k-step
        # launch must not keep writing rows of a bucket that is no longer being gathered)
        env.set_done_bits_target(None)
        k2 = max(args.gather_block, min(400, max(args.steps, 64)))
        leg = timed_leg(env, actions, k2, args.warmup + args.steps)
        physics_ms = leg["avg_launch_us"] * 1e-3 * leg["launches"] / k2
        coll = collective_report(dist, world, rank, dev, gather, local_elapsed * 1e3 / args.steps, collective=collective, physics_ms=physics_ms)

    # the steady state beside the driver's short shape (20 timed launches are 0.5 ms: the clock also holds the first
    # launch's latency and the closing synchronise): the same env, the same ring, HIP events around `--sustained-steps`
    # further launches at the headline population
    sustained = None
    aux_ok = rank == 0 and world == 1 and not multi and args.api == "step"
    if aux_ok and args.sustained_steps > 0:
        leg = timed_leg(env, actions, args.sustained_steps, args.warmup + args.steps)
        us = leg["avg_launch_us"]
        gbs = env.algorithmic_bytes() * n / (us * 1e-6) / 1e9
        sustained = {"launches": leg["launches"], "avg_launch_us": us, "host_enqueue_us": leg["host_enqueue_us"], "host_bound": leg["host_bound"],
                     "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "env_steps_per_s": n / (us * 1e-6),
                     "what": f"{leg['launches']} consecutive env.step() launches at {n} drones after the timed region, HIP events on the launch stream"}
        assert bool(torch.isfinite(env.state).all()), "non-finite state after the sustained leg"

    # the same kernel with its state far outside the 256 MiB Infinity Cache (2^23 drones: 470 MB of state):
    # what a GPU-filling population sees; the 2^20-drone state (59 MB) lives in that cache between steps
    beyond = None
    if aux_ok and not args.no_beyond_mall and venv is None:
        nb = 1 << 23
        big = make_env(nb, False)
        big.reset()
        acts_b = sticks.ema_noise_device(4, nb, dev, seed=99)
        launch_bytes = big.algorithmic_bytes() * nb                        # what one launch of the step kernel moves
        # the box's own streaming ceiling, measured in THIS process around the kernel run: a plain copy that reads and
        # writes the same number of bytes per launch as the step kernel (1.1 GB, nothing survives in the 256 MiB cache),
        # once with 16 bytes per lane (what the guide's 6.29 TB/s "achievable" figure is) and once with the step
        # kernel's own access shape, one dword per lane
        L = _lib_mod.lib()
        cf = (launch_bytes // 8) // 1024 * 1024                            # floats copied per launch: read + write = launch_bytes
        src = torch.empty(cf, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)

        def time_copy(fn, reps=40):
            for _ in range(5):
                _lib_mod.check(fn(dst.data_ptr(), src.data_ptr(), cf, torch.cuda.current_stream().cuda_stream))
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(reps):
                _lib_mod.check(fn(dst.data_ptr(), src.data_ptr(), cf, torch.cuda.current_stream().cuda_stream))
            c1.record()
            torch.cuda.synchronize()
            return 8.0 * cf / (c0.elapsed_time(c1) * 1e-3 / reps) / 1e9      # GB/s, read + write

        leg_preheat(big, acts_b)
        copy_before = time_copy(L.fpv_diag_stream_copy_wide)
        leg = timed_leg(big, acts_b, 100, 0, repeats=3)
        us = leg["avg_launch_us"]
        # the same chain in the plain order (every launch from drone 0): what the rotation of the traversal buys beyond the cache
        rotation = big.rotation
        plain_us = None
        if rotation:
            big.set_rotation(0)
            plain_us = timed_leg(big, acts_b, 100, 300, repeats=3)["avg_launch_us"]
            big.set_rotation(-1)
        copy_after = time_copy(L.fpv_diag_stream_copy_wide)
        copy_dword = time_copy(L.fpv_diag_stream_copy)
        gbs = launch_bytes / (us * 1e-6) / 1e9
        ceiling = max(copy_before, copy_after)
        beyond = {"drones": nb, "state_MB": round(big.state.numel() * 4 / 1e6 + (big.state_h.numel() * 2 / 1e6 if big.state_h is not None else 0)),
                  "avg_launch_us": us, "repeats_us": leg["repeats_us"], "launches_per_repeat": leg["launches"],
                  "host_enqueue_us": leg["host_enqueue_us"], "host_bound": leg["host_bound"],
                  "achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "env_steps_per_s": nb / (us * 1e-6),
                  "copy_ceiling_GBs": ceiling, "frac_of_copy_ceiling": gbs / ceiling,
                  "rotation_drones": rotation, "plain_order_avg_launch_us": plain_us,
                  "rotation_note": (f"each launch starts {rotation} drones before the previous one's start and wraps (fpv_set_rotation, automatic beyond the "
                                    "256 MiB Infinity Cache): it begins on the rows the previous launch wrote last; same results, the plain order timed beside it"
                                    if rotation else None),
                  "addresses": {"state": hex(big.state.data_ptr()), "ld": big.ld, "action": hex(acts_b.data_ptr()),
                                "reward": hex(big.reward.data_ptr()), "done": hex(big.done.data_ptr())},
                  "copy_GBs": {"float4_before": copy_before, "float4_after": copy_after, "dword": copy_dword,
                               "bytes_per_launch": 8 * cf,
                               "what": "fpv_diag_stream_copy_wide / fpv_diag_stream_copy: dst[i] = src[i], read + write bytes equal to one "
                                       "step-kernel launch at 2^23 drones, same process, before and after the kernel run"}}
        del big, acts_b, src, dst

    # what a launch costs as a function of the population, on this box, in this process: the same kernel at three quarters,
    # at one and at twice the headline population - all inside the Infinity Cache and all well beyond the L2s, so that in the
    # plain order every byte comes from ONE cache level (half the population, 2^19 drones, was a point of this line while its
    # row stride made the rows collide in the L2s' sets; with the stride of fpv_recommended_ld the L2s hold it and it runs
    # 2.4 us below the line; fp16 storage writes 39 B per drone: its points start at the headline population) -
    # each leg on fresh buffers after its own
    # time-based preheat, launched from the C loop (fpv_rollout: the same single-step kernel; one Python call per ring span,
    # so the host's per-call cost cannot bound a leg), median of three 400-launch repeats.  The straight line
    # t(n) = floor + bytes / rate says how much of the headline launch is the per-launch floor of a dependent
    # kernel chain and what the rest streams at - when the points sit on a line; fit_launch_time() says when they do not
    fit_legs = None
    if aux_ok and not args.no_beyond_mall and n == (1 << 20) and not args.racer and venv is None:
        fit_legs = []
        for nn in ((n, 3 * n // 2, 2 * n) if args.fp16_state else (3 * n // 4, n, 2 * n)):
            e = make_env(nn, False)
            e.set_rotation(0)       # the plain order: the line is about ONE cache level (the rotation lets the L2s serve a part at 2^20 and 2^21 drones)
            e.reset()
            aa = sticks.ema_noise_device(8, nn, dev, seed=5)
            leg_preheat(e, aa)
            leg = timed_leg(e, aa, 400, 0, repeats=3, c_loop=True)
            leg.update(drones=nn, bytes_per_launch=e.algorithmic_bytes() * nn)
            fit_legs.append(leg)
            del e, aa

    if rank == 0:
        state_bytes = env.algorithmic_bytes()                                # 133 B fp32 / 89 B fp16 state
        kernel_s = dev_ms * 1e-3 / n_launches                                # avg launch-to-launch on the stream
        steps_per_launch = args.steps / n_launches
        if args.api == "rollout":
            # k-step kernel: the state (and nothing else) is amortised over the k steps of a launch
            io = 16 + 0                                                      # action read; reward/done only after the last step
            rw_state = state_bytes - 21
            bytes_per_step = io + (rw_state + 5) / steps_per_launch
        else:
            bytes_per_step = state_bytes
        rf = hbm_roofline(n * args.steps / elapsed, bytes_per_step, n, steps_per_launch, kernel_s)
        achieved = rf["achieved"]
        traffic, traffic_src = None, None
        tp = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.isfile(tp) and not args.fp16_state and not args.racer and args.api == "step" and n == (1 << 20):
            tj = json.load(open(tp))
            if measurement_is_current(tj):
                traffic, traffic_src = tj.get("hbm_bytes_per_launch"), tj.get("source")
            else:
                traffic_src = "stale: profiles/pmc_traffic.json was measured on another build of the kernels (re-run tools/pmc_probe.py)"
        kernel = ("fpv_racer_step_kernel" if args.racer else "fpv_drone_step_h_kernel" if args.fp16_state else "fpv_drone_step_kernel")
        if args.api == "rollout":
            kernel = kernel.replace("_step_", "_rollout_")
        cfg_name = ("Racer.step (" + args.racer + "), " if args.racer else "configs[3]: fp16 state / fp32 integrator, " if args.fp16_state else "configs[2]: ")
        out = {
            "metric": "env-steps/sec at N=1.05M drones, dt=1ms; 1/2/4/8 GPU + CPU ref",
            "value": n * world * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "rehearsal (every rank on GPU 0, gloo): not a measurement" if args.rehearse_on_one_gpu else "synthetic",
            "state_storage": ("eleven 16-bit words per drone: f16 v with 5-bit low words, smallest-three 15-bit fixed-point q, f16 rates and thrust; f32 p"
                              if args.fp16_state else "f32"),
            "config": {"workload": cfg_name
                       + f"{n} drones/GPU, EMA-noise sticks (noise_smooth_test profile), fp32 math, dt=1ms, "
                       + ("no auto-reset" if args.no_auto_reset else f"in-kernel auto-reset on ground contact or |z|>{args.ceiling:g} m"),
                       "drones_per_gpu": n, "global_drones": n * world, "action_ring": ring, "api": args.api,
                       "partitions": (venv.partitions if venv is not None else 1),
                       "partition_streams": (venv.stream_report if venv is not None else None),
                       "steps_per_launch": steps_per_launch, "rotation_drones": env.rotation,
                       "parallelism": f"shard{world}" + ("+allgather(done_bits x" + str(args.gather_block) + " steps"
                                                              + (", last_return" if args.gather_returns else "") + ")" if gather is not None else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": rf["frac"], "achieved_events": rf["achieved_events"], "frac_events": rf["frac_events"],
                         "frac_formula": "frac = value / n_gpus x algorithmic_bytes_per_env_step / 1e9 / peak (the wall clock of the whole timed region, synchronise tail "
                                         "included - the same clock as `value` and `ms_per_step`); frac_events = algorithmic bytes per launch / avg_launch_us / peak (HIP events, "
                                         "first launch to last launch - the clock of the rocprofv3 kernel trace)",
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_note": ("the counter passes serialise the launches and the L2s do not keep their lines from one to the next there (profiles/pmc_traffic.json: "
                                          "pmc_pass_step_kernel_avg_ns against kernel_trace_avg_ns): `traffic` is what a launch with COLD L2s asks the memory side for - its "
                                          "algorithmic bytes; in the running chain the rotation lets the L2s answer a part of the state reads, which no counter pass can watch "
                                          "without undoing it" if traffic is not None else None),
                         "kernel": kernel, "algorithmic_bytes_per_env_step": bytes_per_step,
                         "avg_launch_us": kernel_s * 1e6,
                         "host_enqueue_us": host_enqueue_s * 1e6 / max(k_host, 1) * steps_per_launch,
                         "host_enqueue_over": f"the first {k_host} steps of the timed region (before the hardware queue can fill)",
                         "host_bound": bool(host_enqueue_s / max(k_host, 1) > 0.9 * dev_ms * 1e-3 / args.steps),
                         "cache_note": "this line is a STEP-ONLY chain (sticks from a pre-generated ring: BASELINE configs[2] is a noise profile, not a policy). At 2^20 drones "
                                       "the 59 MB state is re-read from the 256 MiB Infinity Cache (MALL) every step - and, with the rotation of the "
                                       "traversal (config.rotation_drones: each launch starts on the rows the previous one wrote last), about half of it from the L2s; "
                                       "only the action stream and reward/done cross HBM - `beyond_mall` is the same kernel at 2^23 drones, where the traversal rotates so that "
                                       "each launch starts on what the cache still holds. With another kernel's pass over the state between two steps (a policy: action -> step -> "
                                       "action) the caches hold what THAT kernel touched last and the rotation is neutral - measured: profiles/r06_closed_loop.md (step kernel "
                                       "21.6-21.7 us inside the loop with or without it at 2^20 drones)",
                         "frac_beyond_mall": beyond["frac"] if beyond else None, "beyond_mall": beyond,
                         "sustained": sustained,
                         "launch_time_fit": None,
                         "note": "avg_launch_us = HIP-event time over the timed region / launches (includes inter-launch gaps); achieved / frac follow `value` "
                                 "(wall clock), achieved_events / frac_events follow avg_launch_us - see frac_formula"},
        }
        if world == 1 and args.api == "step":
            out["roofline"]["xcd_map"] = xcd_map_probe(dev, int(_lib_step_grid(n)))
        if venv is not None:
            out["roofline"]["note"] = (f"split phase: one step = {venv.partitions} launches on {venv.partitions} streams that overlap; avg_launch_us is the "
                                       "time per full-population step, achieved = 133 B x n / that; " + out["roofline"]["note"])
        if fit_legs is not None:
            lf = fit_launch_time([(g_["drones"], g_["bytes_per_launch"], g_["avg_launch_us"]) for g_ in fit_legs])
            head_us = sustained["avg_launch_us"] if sustained else kernel_s * 1e6
            lf.update(legs=[{k_: g_[k_] for k_ in ("drones", "avg_launch_us", "repeats_us", "launches", "host_enqueue_us", "host_bound")} for g_ in fit_legs],
                      launched_by="fpv_rollout (k single-step launches per C call) in the PLAIN order (fpv_set_rotation 0), fresh buffers, time-based preheat per leg, median of 3 x 400 launches",
                      headline_point="roofline.sustained" if sustained else "the timed region",
                      floor_share_of_headline_launch=(lf["floor_us"] / head_us) if lf["valid"] else None)
            if any(g_["host_bound"] for g_ in fit_legs) and lf["valid"]:
                lf.update(valid=False, invalid_reason="a leg's host enqueue time is within 10 % of its launch time: the host's launch loop, not the GPU, set that point")
            out["roofline"]["launch_time_fit"] = lf
        if coll is not None:
            out["collective"] = coll
        if args.api == "rollout":
            # the k-step kernel is limited by instruction issue, not HBM: price it against the vector-ALU issue peak.
            # VALU instructions per wave come from a counter-only rocprofv3 pass (tools/pmc_valu.py ->
            # profiles/pmc_valu.json) and are used ONLY for the configuration they were measured on - same kernel sources
            # (hash), same kernel family, same steps per launch, quiet steps (no per-step mask rows) - otherwise the line
            # says so instead of quoting a stale count
            out["roofline"]["hbm_view"] = {k: out["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "achieved_events", "frac_events")}
            valu, why = None, None
            vp = os.path.join(REPO, "profiles", "pmc_valu.json")
            fam = "racer" if args.racer else "fp16" if args.fp16_state else "f32"
            if not os.path.isfile(vp):
                why = "profiles/pmc_valu.json is missing (run tools/pmc_valu.py on the GPU box)"
            else:
                vj = json.load(open(vp))
                ent = vj.get("kernels", {}).get(fam)
                if not measurement_is_current(vj):
                    why = "stale: profiles/pmc_valu.json was measured on another build of the kernels (re-run tools/pmc_valu.py)"
                elif ent is None or gather is not None or abs(steps_per_launch - ent["steps_per_launch"]) > 0.5 or n != vj.get("drones"):
                    # (a timed region that starts in the middle of the action ring has one or two shorter launches: the
                    # average steps per launch may sit a fraction below the ring span the count was taken at)
                    why = "profiles/pmc_valu.json holds no count for this configuration (kernel family / steps per launch / per-step mask rows)"
                else:
                    inst = ent["valu_per_wave"] / ent["steps_per_launch"]
                    rate = inst * n * args.steps / elapsed / 1e9                          # follows `value` (wall clock), like the HBM view
                    rate_ev = inst * n * steps_per_launch / kernel_s / 1e9                # HIP events
                    valu = {"valu_inst_per_env_step": inst, "salu_inst_per_env_step": ent.get("salu_per_wave", 0) / ent["steps_per_launch"],
                            "source": vj.get("source"), "kernel": ent.get("kernel"),
                            "achieved_Glane_inst_per_s": rate, "peak_Glane_inst_per_s": VALU_PEAK_GINST, "frac": rate / VALU_PEAK_GINST,
                            "achieved_events_Glane_inst_per_s": rate_ev, "frac_events": rate_ev / VALU_PEAK_GINST}
            out["roofline"]["valu"] = valu
            out["roofline"]["valu_unavailable"] = why
            if valu is not None:
                # priced against the issue peak only with a counted instruction mix for exactly this configuration; without
                # one the line keeps bound = "hbm" and its algorithmic-bytes figures (an HBM fraction labelled as a VALU
                # bound would be a number about nothing)
                out["roofline"].update(bound="valu", achieved=valu["achieved_Glane_inst_per_s"], peak=VALU_PEAK_GINST,
                                       unit="G lane-instructions/s", frac=valu["frac"], achieved_events=valu["achieved_events_Glane_inst_per_s"],
                                       frac_events=valu["frac_events"],
                                       frac_formula="frac = value / n_gpus x valu.valu_inst_per_env_step / 1e9 / peak (wall clock, the clock of `value`); frac_events: the same "
                                                    "instructions over avg_launch_us (HIP events)")
                out["roofline"]["note"] = ("bound = vector-ALU issue (256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T lane-instructions/s); "
                                           "`hbm_view` keeps the algorithmic-bytes figures of the same run; " + out["roofline"]["note"])
            else:
                out["roofline"]["note"] = ("the k-step kernel is bound by vector-ALU issue, but no instruction count is on file for this "
                                           "configuration (valu_unavailable): the figures shown are the algorithmic-bytes (HBM) view; " + out["roofline"]["note"])
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(params)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
