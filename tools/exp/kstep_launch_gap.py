#!/usr/bin/env python3
"""Per-launch overhead of the k-step kernel: the same number of env-steps as launches of k = 8 .. 1024 steps (a ring of
32 action batches re-read, stride 0 between ring passes is not possible, so the ring is as long as the largest k)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fpyv_amd import load_params, sticks
from fpyv_amd.env import DroneBatch
dev = torch.device("cuda:0")
n = 1 << 20
env = DroneBatch(load_params(fps=1000, ceiling=100.0), n, device=dev, auto_reset=True, with_accel=False)
env.reset()
kmax = 256
acts = sticks.ema_noise_device(kmax, n, dev, seed=1)          # 4.3 GB of sticks
total = 2048
for k in (8, 16, 32, 64, 128, 256):
    for rep in range(2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for j in range(total // k):
            r0 = (j * k) % kmax
            env.rollout(acts[r0:r0 + k])
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    print(f"k = {k:4d}: {us / total:6.3f} us per env-step   {us / (total // k):8.1f} us per launch", flush=True)
