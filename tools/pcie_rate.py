#!/usr/bin/env python3
"""What the headline would be if the sticks came from the HOST every step ("PCIe note"): the boundary takes
device pointers, so this is not `value` - it is the measured price of a caller that produces actions on the CPU.
Pinned host batches [n, 4] fp32 are copied to the device on a copy stream, double-buffered, while the previous step
runs; wall clock over K steps.

    python tools/pcie_rate.py [--drones 1048576] [--steps 300]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--drones", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=300)
a = ap.parse_args()
dev = torch.device("cuda:0")
n = a.drones
main = torch.cuda.Stream(device=dev)
copy = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(main)
env = DroneBatch(load_params(fps=1000, ceiling=100.0), n, device=dev, auto_reset=True, with_accel=False)
env.reset()
host = [sticks.ema_noise_device(1, n, dev, seed=s)[0].cpu().pin_memory() for s in range(4)]     # four distinct pinned batches
devb = [torch.empty((n, 4), dtype=torch.float32, device=dev) for _ in range(2)]
ready = [torch.cuda.Event() for _ in range(2)]
used = [torch.cuda.Event() for _ in range(2)]


def run(k):
    for t in range(k):
        b = t & 1
        with torch.cuda.stream(copy):
            copy.wait_event(used[b])                      # the step that read this buffer two steps ago has finished
            devb[b].copy_(host[t & 3], non_blocking=True)
            ready[b].record(copy)
        main.wait_event(ready[b])
        env.step(devb[b], return_imu=False)
        used[b].record(main)


for e in used:
    e.record(main)
run(20)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(a.steps)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
mb = n * 16 / 1e6
print(f"{n} drones, sticks uploaded from pinned host memory every step ({mb:.1f} MB), copy overlapped with the previous step: "
      f"{dt * 1e6:.1f} us per step = {n / dt / 1e9:.2f} G env-steps/s; the upload alone moves {mb / dt / 1e3:.1f} GB/s over PCIe")
# the copy alone
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(50):
    devb[0].copy_(host[t & 3], non_blocking=True)
torch.cuda.synchronize()
dc = (time.perf_counter() - t0) / 50
print(f"the upload alone: {dc * 1e6:.1f} us per batch = {mb / dc / 1e3:.1f} GB/s; the step kernel alone ~22 us")
