#!/usr/bin/env python3
"""Closed loop without a single transpose or copy kernel: the SoA state IS the observation matrix
obs[13, N]; a policy's `W @ obs` produces sticks [4, N], which fpv_step consumes in place
(fpv_buffers_t.action_ld).  Prints env-steps/s of policy + physics together.

    python examples/closed_loop_policy.py --drones 1048576 --steps 2000
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpyv_amd import load_params  # noqa: E402
from fpyv_amd.env import FpvVecEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--drones", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--hidden", type=int, default=0, help="0: linear policy; >0: one hidden layer of this width")
a = ap.parse_args()
dev = torch.device("cuda:0")
env = FpvVecEnv(load_params(fps=1000, ceiling=100.0), num_envs=a.drones, device=dev, track_episodes=False)
env.reset()
n = env.num_envs
torch.manual_seed(0)
if a.hidden:
    W1, W2 = torch.randn(a.hidden, 13, device=dev) * 0.1, torch.randn(4, a.hidden, device=dev) * 0.1
    policy = lambda x: torch.tanh(W2 @ torch.relu(W1 @ x))        # noqa: E731
else:
    W = torch.randn(4, 13, device=dev) * 0.02
    policy = lambda x: torch.tanh(W @ x)                          # noqa: E731
obs_soa = env.batch.state[:13, :n]                                # [13, N] view of the live state
for _ in range(50):
    env.batch.step(policy(obs_soa), return_imu=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    env.batch.step(policy(obs_soa), return_imu=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n} drones, {a.steps} closed-loop steps ({'MLP ' + str(a.hidden) if a.hidden else 'linear'} policy + physics): "
      f"{dt / a.steps * 1e6:.1f} us per step = {n * a.steps / dt / 1e9:.2f} G env-steps/s")
