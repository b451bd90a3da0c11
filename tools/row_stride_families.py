#!/usr/bin/env python3
"""The row stride against the L2s' sets for the kernel families that keep MORE rows per drone than the plain kernel's 14 (Racer:
20 - 29 rows of one matrix; Kahan rows, accel rows: further matrices of the same stride).  For each family at --n drones: the
launch time through env.step with the automatic rotation for the eight 256-byte classes of stride (ld = a multiple of 512
floats + c * 64) and for fpv_recommended_ld(n).  One batch per stride (a batch allocates with the stride it is given).

    python tools/row_stride_families.py [--n 1048576] [--families racerW racerD ...]      # GPU box"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fpyv_amd import _lib, load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch, RacerBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--families", nargs="*", default=["f32", "kahan", "accel", "aos", "h", "racerD", "racerW", "racerWC"])
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--launches", type=int, default=320)
a = ap.parse_args()
dev = torch.device("cuda:0")
p = load_params(fps=1000, ceiling=100.0)
ring = 32 if a.n <= (1 << 21) else 4
acts = sticks.ema_noise_device(ring, a.n, dev)
pid = np.array([[0.004, 0.02, 1e-6], [0.003, 0.01, 2e-6], [0.002, 0.005, 0.0]])
racer = {"racerW": dict(racer_omega_dt=False), "racerD": dict(racer_omega_dt=True),
         "racerWC": dict(racer_omega_dt=False, racer_pid_variant=1, racer_pid=-pid, pid_integral_clip=0.05, pid_min_output=-0.004, pid_max_output=0.006,
                         pid_derivative_transition_rate=0.3)}
drone = {"f32": {}, "accel": dict(with_accel=True), "kahan": dict(kahan_position=True), "aos": dict(with_obs_aos=True), "h": dict(fp16_state=True)}
L = _lib.lib()
recommended = L.fpv_recommended_ld          # the real one; a batch asks `L.fpv_recommended_ld(n)` for its stride: shadowed per build below


def build(f, ld):
    L.fpv_recommended_ld = lambda n: ld
    try:
        if f in racer:
            e = RacerBatch(p.replace(**dict(dict(mode=1, racer_pid=pid, ceiling=50.0), **racer[f])), a.n, device=dev, auto_reset=True)
        else:
            e = DroneBatch(p, a.n, device=dev, auto_reset=True, **dict(dict(with_accel=False), **drone[f]))
    finally:
        L.fpv_recommended_ld = recommended
    assert e.ld == ld
    e.reset()
    return e


ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def run(e, launches):
    torch.cuda.synchronize()
    ev0.record()
    for t in range(launches):
        e.step(acts[t % ring], return_imu=False)
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e3 / launches


rec = int(recommended(a.n))
base = (a.n + 511) // 512 * 512
lds = [base + c * 64 for c in range(8)] + [rec]
for f in a.families:
    res = {}
    for ld in lds:
        e = build(f, ld)
        run(e, 64)
        res[ld] = statistics.median(run(e, a.launches) for _ in range(a.rounds))
        rot = e.rotation // 128
        del e
        torch.cuda.empty_cache()
    print(f"{f:8s} n={a.n} rotation {rot:5d} blocks  classes 0..7: " + " ".join(f"{res[ld]:6.2f}" for ld in lds[:8]) + f"   recommended (n+{rec - a.n}, class {rec % 512 // 64}): {res[rec]:6.2f}", flush=True)
