#!/usr/bin/env python3
"""What binary16 sticks (fpv_buffers_t.action_f16: a half-precision policy's output consumed as it is) buy: the headline
workload with the action ring given as float16 against the same ring as float32 - single-step launches (125 against 133
algorithmic bytes per env-step) and the k-step kernel (8 + 117/k against 16 + 117/k), interleaved in one process.

    python tools/f16_sticks_rate.py [--drones 1048576]
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--drones", type=int, default=1 << 20)
ap.add_argument("--rounds", type=int, default=9)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
n = a.drones
p = load_params(fps=1000, ceiling=100.0)
f32 = sticks.ema_noise_device(32, n, dev, seed=1234)
f16 = f32.to(torch.float16)
env = DroneBatch(p, n, device=dev, auto_reset=True, with_accel=False)
env.reset()
res = {}
for r in range(a.rounds):
    for name, acts in (("float32 sticks", f32), ("float16 sticks", f16)):
        for api, fused, reps in (("single-step launches", False, 8), ("k-step kernel (k = 32)", True, 48)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                env.rollout(acts, fused=fused)
            e1.record()
            torch.cuda.synchronize()
            if r:
                res.setdefault((api, name), []).append(e0.elapsed_time(e1) * 1e3 / (reps * 32))
for (api, name), v in sorted(res.items()):
    med = statistics.median(v)
    print(f"{api:26s} {name}: median {med:7.3f} us per env-step  min {min(v):7.3f}   {n / med / 1e3:7.2f} G env-steps/s", flush=True)
