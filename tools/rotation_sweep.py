#!/usr/bin/env python3
"""Where is the optimum of the rotation of the traversal for every kernel family?  One process, the same buffers, interleaved
rounds: the launch time at explicit rotations around the automatic one (fpv_set_rotation) for each family at --n drones.

    python tools/rotation_sweep.py [--n 1048576] [--families f32 kahan racerW ...] [--rounds 5]

One line per family: automatic rotation in blocks of 128 drones, then `blocks:us` pairs."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch, RacerBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--families", nargs="*", default=["f32", "accel", "noise", "kahan", "aos", "h", "racerW", "racerD", "racerWC"])
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--launches", type=int, default=320)
ap.add_argument("--base-blocks", type=int, default=0, help="sweep around this many blocks instead of the automatic rotation (a population the rule leaves in the plain order)")
ap.add_argument("--scales", nargs="*", type=float, default=[0.0, 0.6, 0.7, 0.8, 0.85, 0.9, 0.95, 1.0, 1.05, 1.1, 1.2])
a = ap.parse_args()
dev = torch.device("cuda:0")
p = load_params(fps=1000, ceiling=100.0)
ring = 32 if a.n <= (1 << 21) else 4
acts = sticks.ema_noise_device(ring, a.n, dev)
pid = np.array([[0.004, 0.02, 1e-6], [0.003, 0.01, 2e-6], [0.002, 0.005, 0.0]])
racer = {"racerW": dict(racer_omega_dt=False), "racerD": dict(racer_omega_dt=True),
         "racerWC": dict(racer_omega_dt=False, racer_pid_variant=1, racer_pid=-pid, pid_integral_clip=0.05, pid_min_output=-0.004, pid_max_output=0.006,
                         pid_derivative_transition_rate=0.3)}
drone = {"f32": {}, "accel": dict(with_accel=True), "noise": dict(stick_noise=True, noise_seed=1), "kahan": dict(kahan_position=True), "aos": dict(with_obs_aos=True),
         "h": dict(fp16_state=True)}


def build(f):
    if f in racer:
        e = RacerBatch(p.replace(**dict(dict(mode=1, racer_pid=pid, ceiling=50.0), **racer[f])), a.n, device=dev, auto_reset=True)
    else:
        e = DroneBatch(p, a.n, device=dev, auto_reset=True, **dict(dict(with_accel=False), **drone[f]))
    e.reset()
    return e


ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def run(e, f, launches):
    torch.cuda.synchronize()
    ev0.record()
    for t in range(launches):
        e.step(None if f == "noise" else acts[t % ring], return_imu=False)
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e3 / launches


for f in a.families:
    e = build(f)
    run(e, f, 64)
    auto = a.base_blocks or e.rotation // 128
    rots = sorted({int(auto * s) // 8 * 8 for s in a.scales}) if auto else [0]
    res = {r: [] for r in rots}
    for k in range(a.rounds + 1):
        for r in rots:
            e.set_rotation(r * 128)
            run(e, f, 32)                 # the chain settles into the new rotation
            t = run(e, f, a.launches)
            if k:
                res[r].append(t)
    best = min(rots, key=lambda r: statistics.median(res[r]))
    print(f"{f:8s} n={a.n} auto {auto:6d} blocks  best {best:6d} ({best / auto if auto else 0:.2f} of auto)   " +
          "  ".join(f"{r}:{statistics.median(res[r]):.2f}" for r in rots), flush=True)
    del e
    torch.cuda.empty_cache()
