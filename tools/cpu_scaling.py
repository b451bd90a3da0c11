#!/usr/bin/env python3
"""Thread-scaling of the CPU baseline (the float64 C oracle, native build) on this host: env-steps/s at
1, 2, 4, ... threads, with what the OS says about the CPUs this job may use.  No GPU needed."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from fpyv_amd import load_params  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    p = load_params(fps=1000)
    oracle.lib(native=True)
    n, T = 1 << 17, 16
    acts = np.random.default_rng(0).standard_normal((T, n, 4)) * 0.2
    info = {"os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "usable_cpus": bench.usable_cpus(),
            "omp_max_threads": oracle.max_threads(), "nproc": subprocess.run(["nproc"], capture_output=True, text=True).stdout.strip()}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.isfile(path):
            info[path] = open(path).read().strip()
    try:
        info["cpu_model"] = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    rows = []
    th = 1
    while th <= max(1, min(256, os.cpu_count() or 1)):
        st = oracle.drone_initial_state(n, p.init_position, p.init_velocity, [0, 0, 0])
        oracle.drone_run(p, st, acts[:2], threads=th, native=True)
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            oracle.drone_run(p, st, acts, threads=th, native=True)
            reps += 1
        rate = n * T * reps / (time.perf_counter() - t0)
        rows.append({"threads": th, "env_steps_per_s": rate})
        print(f"threads {th:4d}: {rate / 1e6:9.2f} M env-steps/s  ({rate / rows[0]['env_steps_per_s']:.1f}x)", flush=True)
        th *= 2
    print(json.dumps({"host": info, "scaling": rows}))


if __name__ == "__main__":
    main()
