#!/usr/bin/env python3
"""What a K = 20 timed region (the driver's bench shape) pays on top of the kernels: wall clock around
[K env.step launches + end-of-region sync] against the HIP-event time of the same launches, for three ways
of waiting for the GPU: torch.cuda.synchronize() (hipDeviceSynchronize), a host spin on stream.query()
followed by synchronize(), and hipStreamSynchronize on the launch stream.  Run once plain and once with
ROC_ACTIVE_WAIT_TIMEOUT set, to see what the runtime's own wait policy costs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda:0")
n, K = 1 << 20, int(os.environ.get("K", "20"))
env = DroneBatch(load_params(fps=1000, ceiling=100.0), n, device=dev, auto_reset=True, with_accel=False)
env.reset()
acts = sticks.ema_noise_device(32, n, dev, seed=1)
for _ in range(20000):                       # leave the idle clocks
    env.step(acts[0], return_imu=False)
torch.cuda.synchronize()
stream = torch.cuda.current_stream()


def wait_sync():
    torch.cuda.synchronize()


def wait_spin():
    while not stream.query():
        pass
    torch.cuda.synchronize()


def wait_stream():
    stream.synchronize()


print("ROC_ACTIVE_WAIT_TIMEOUT =", os.environ.get("ROC_ACTIVE_WAIT_TIMEOUT"), " HSA_ENABLE_INTERRUPT =", os.environ.get("HSA_ENABLE_INTERRUPT"))
for name, wait in (("synchronize", wait_sync), ("spin+synchronize", wait_spin), ("stream.synchronize", wait_stream)):
    walls, devs = [], []
    for rep in range(30):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for t in range(K):
            env.step(acts[t % 32], return_imu=False)
        e1.record()
        t_launch = time.perf_counter()
        wait()
        t1 = time.perf_counter()
        walls.append((t1 - t0) * 1e6)
        devs.append(e0.elapsed_time(e1) * 1e3)
    walls.sort(); devs.sort()
    print(f"{name:20s} K={K}: wall median {walls[15]:7.1f} us  min {walls[0]:7.1f}  device median {devs[15]:7.1f} us  "
          f"-> overhead {walls[15] - devs[15]:5.1f} us; per step {walls[15] / K:.2f} us; host launch part {(t_launch - t0) * 1e6:.1f} us")
