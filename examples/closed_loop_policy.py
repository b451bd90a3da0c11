#!/usr/bin/env python3
"""Closed loop without a single transpose or copy kernel, and - with --partitions 2 - without the GPU ever idling at a
kernel boundary: the SoA state IS the observation matrix obs[13, N]; a policy's `W @ obs` produces sticks [4, N], which
fpv_step consumes in place (fpv_buffers_t.action_ld).

Split phase (FpvVecEnv.step_async / step_wait, the gym VectorEnv convention per partition): the population is cut into
column partitions of the same tensors, each with its own kernel chain on its own stream.  The policy of partition A
runs while partition B steps:

    for part in range(env.partitions):
        with torch.cuda.stream(env.stream(part)):          # this partition's chain: policy -> step -> policy -> ...
            obs, reward, done, info = env.step_wait(part, sync=False)
            env.step_async(part, policy(obs_soa_view[part]))

Prints env-steps/s of policy + physics together, for the unpartitioned loop and for the split-phase loop.

    python examples/closed_loop_policy.py --drones 1048576 --steps 2000 --hidden 64
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
ap.add_argument("--partitions", type=int, nargs="+", default=[1, 2], help="partition counts to run, one after the other")
ap.add_argument("--rotation", type=int, default=-1, help="fpv_set_rotation: -1 automatic, 0 plain order, > 0 drones (unpartitioned loop only)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
torch.manual_seed(0)
if a.hidden:
    W1, W2 = torch.randn(a.hidden, 13, device=dev) * 0.1, torch.randn(4, a.hidden, device=dev) * 0.1
    policy = lambda x: torch.tanh(W2 @ torch.relu(W1 @ x))        # noqa: E731
else:
    W = torch.randn(4, 13, device=dev) * 0.02
    policy = lambda x: torch.tanh(W @ x)                          # noqa: E731
name = f"MLP 13-{a.hidden}-4" if a.hidden else "linear 13-4"
results = {}
for parts in a.partitions:
    env = FpvVecEnv(load_params(fps=1000, ceiling=100.0), num_envs=a.drones, device=dev, track_episodes=False, partitions=parts)
    env.reset()
    n = env.num_envs
    if env.partitions == 1:
        obs_soa = env.batch.state[:13, :n]                        # [13, N] view of the live state
        env.batch.set_rotation(a.rotation)

        def run(k):
            for _ in range(k):
                env.batch.step(policy(obs_soa), return_imu=False)
    else:
        views = [env.batch.state[:13, lo:hi] for lo, hi in (env.partition_range(p) for p in range(env.partitions))]
        streams = [env.stream(p) for p in range(env.partitions)]

        def run(k):
            for _ in range(k):
                for p in range(env.partitions):
                    with torch.cuda.stream(streams[p]):           # policy and step of one partition: one chain, one stream
                        env.step_async(p, policy(views[p]))
            for p in range(env.partitions):
                env.step_wait(p)
    run(50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    results[env.partitions] = dt / a.steps * 1e6
    print(f"{n} drones, {a.steps} closed-loop steps, {name} policy + physics, {env.partitions} partition(s)"
          + (f" (streams: {env.stream_report})" if env.stream_report else "")
          + f": {dt / a.steps * 1e6:.1f} us per step = {n * a.steps / dt / 1e9:.2f} G env-steps/s", flush=True)
    assert bool(torch.isfinite(env.batch.state).all())
    env.close()
if 1 in results and len(results) > 1:
    for p, us in results.items():
        if p != 1:
            print(f"split phase with {p} partitions: {100 * (results[1] / us - 1):+.1f} % steps per second against the unpartitioned loop")
