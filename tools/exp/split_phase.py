#!/usr/bin/env python3
"""A/B in ONE process on one box: the 2^20-drone population stepped (a) as one batch through env.step, (b) as P column
partitions through FpvVecEnv.step_async (P independent kernel chains on P streams, never joined), (c) the same with a
closed loop: a linear policy on each partition's observation view between step_wait and step_async, (d) the closed loop
on the unpartitioned env.  Wall clock between synchronises, medians over repetitions.

    python tools/exp/split_phase.py [--drones N] [--steps K] [--reps R]
"""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import FpvVecEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--drones", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--reps", type=int, default=7)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
n, K = a.drones, a.steps
params = load_params(fps=1000, ceiling=100.0)
ring = sticks.ema_noise_device(32, n, dev, seed=1234)
torch.manual_seed(0)
W = torch.randn(4, 13, device=dev) * 0.02


def timed(fn):
    fn(50)
    torch.cuda.synchronize()
    ts, hs = [], []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        fn(K)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e6)
        hs.append((t1 - t0) / K * 1e6)
    return statistics.median(ts), min(ts), statistics.median(hs)


def report(name, med, mn, host):
    print(f"{name:<72s}: median {med:7.2f} us per step   min {mn:7.2f}   {n / med / 1e3:6.2f} G env-steps/s   host issue {host:6.2f} us per step", flush=True)


def open_loop(parts):
    env = FpvVecEnv(params, num_envs=n, device=dev, track_episodes=False, partitions=parts)
    env.reset()
    if parts == 1:
        def fn(k):
            for t in range(k):
                env.batch._step_raw(ring[t % 32])
    else:
        rng = [env.partition_range(p) for p in range(env.partitions)]
        sl = [[ring[r][lo:hi] for lo, hi in rng] for r in range(32)]

        def fn(k):
            for t in range(k):
                row = sl[t % 32]
                for p in range(env.partitions):
                    env.step_async(p, row[p], ready=True)
            for p in range(env.partitions):
                env.step_wait(p)
    return env, fn


def closed_loop(parts, own_streams):
    env = FpvVecEnv(params, num_envs=n, device=dev, track_episodes=False, partitions=parts)
    env.reset()
    if parts == 1:
        obs = env.batch.state[:13, :n]

        def fn(k):
            for _ in range(k):
                env.batch._step_raw(torch.tanh(W @ obs))
    else:
        views = [env.batch.state[:13, lo:hi] for lo, hi in (env.partition_range(p) for p in range(env.partitions))]
        if own_streams:
            def fn(k):        # the policy of a partition runs on that partition's own stream: two fully independent chains
                for _ in range(k):
                    for p in range(env.partitions):
                        with torch.cuda.stream(env.stream(p)):
                            env.step_async(p, torch.tanh(W @ views[p]))
                for p in range(env.partitions):
                    env.step_wait(p)
        else:
            def fn(k):        # the policy on the caller's stream: step_wait / step_async order it with the partition's chain
                for _ in range(k):
                    for p in range(env.partitions):
                        env.step_wait(p)
                        env.step_async(p, torch.tanh(W @ views[p]))
                for p in range(env.partitions):
                    env.step_wait(p)
    return env, fn


print(f"# tools/exp/split_phase.py: {n} drones, {K} steps per repetition, {a.reps} repetitions, wall clock between synchronises")
# 1 against 2 partitions, repetitions interleaved (same minutes, same clocks)
e1, f1 = open_loop(1)
e2, f2 = open_loop(2)
f1(200); f2(200); torch.cuda.synchronize()
r1, r2 = [], []
for _ in range(a.reps * 2):
    for f, r in ((f1, r1), (f2, r2)):
        t0 = time.perf_counter()
        f(K)
        torch.cuda.synchronize()
        r.append((time.perf_counter() - t0) / K * 1e6)
print(f"interleaved A/B, open loop: 1 partition median {statistics.median(r1):.2f} (min {min(r1):.2f}) us per step; 2 partitions median "
      f"{statistics.median(r2):.2f} (min {min(r2):.2f}) us per step: {100 * (statistics.median(r1) / statistics.median(r2) - 1):+.1f} %", flush=True)
e1.close(); e2.close()
for parts in (1, 2, 3, 4):
    env, fn = open_loop(parts)
    report(f"open loop (ring of sticks), {parts} partition(s)", *timed(fn))
    env.close()
for parts, own in ((1, False), (2, False), (2, True), (3, True)):
    env, fn = closed_loop(parts, own)
    report(f"closed loop tanh(W @ obs), {parts} partition(s)" + (", policy on the partition's stream" if own else ", policy on the caller's stream" if parts > 1 else ""), *timed(fn))
    env.close()
