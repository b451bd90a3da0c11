#!/usr/bin/env python3
"""Soak of the round-4 paths on one MI355X: 200 000 steps of the re-encoded fp16 state at 2^20 drones (finite, unit quaternions after
decoding), then 20 000 steps of the split-phase env (two partitions, auto-reset, episode bookkeeping) against the single batch, bit for bit.

    python tools/soak_fp16_split.py
"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpyv_amd import load_params, sticks
from fpyv_amd.env import DroneBatch, FpvVecEnv
dev = torch.device("cuda:0"); n = 1 << 20
p = load_params(fps=1000, ceiling=100.0)
acts = sticks.ema_noise_device(32, n, dev, seed=3) * 1.5
e = DroneBatch(p, n, device=dev, fp16_state=True, auto_reset=True, track_episodes=True, with_accel=False, rounding_seed=9)
e.reset()
t0 = time.perf_counter()
for it in range(6250):          # 200 000 steps
    e.rollout(acts)
    if it % 1250 == 1249:
        torch.cuda.synchronize()
        s = e.rows_f32(0, 14)
        qn = s[:, 6:10].norm(dim=1)
        print(f"{(it + 1) * 32} steps {time.perf_counter() - t0:5.1f} s finite={bool(torch.isfinite(s).all())} max||q|-1|={float((qn - 1).abs().max()):.2e} |z|max={float(s[:, 2].abs().max()):.1f} mean ep len={float(e.last_length.float().mean()):.0f}", flush=True)
        assert bool(torch.isfinite(s).all()) and float((qn - 1).abs().max()) < 1e-5
# split phase soak: 20 000 closed-loop-shaped steps on 2 partitions with auto-reset against the single batch
v1 = FpvVecEnv(p.replace(ceiling=12.0), num_envs=1 << 18, device=dev, track_episodes=True, with_done_bits=True)
v2 = FpvVecEnv(p.replace(ceiling=12.0), num_envs=1 << 18, device=dev, track_episodes=True, with_done_bits=True, partitions=2)
v1.reset(); v2.reset()
a18 = acts[:, : 1 << 18].contiguous()
for t in range(20000):
    v1.step(a18[t % 32])
    for k in range(2):
        lo, hi = v2.partition_range(k)
        v2.step_async(k, a18[t % 32][lo:hi], ready=True)
for k in range(2):
    v2.step_wait(k)
torch.cuda.synchronize()
print("split-phase soak: state equal", bool(torch.equal(v1.batch.state, v2.batch.state)), "episodes", int(v1.batch.last_length.gt(0).sum()), "equal bookkeeping", bool(torch.equal(v1.batch.last_length, v2.batch.last_length)))
assert torch.equal(v1.batch.state, v2.batch.state) and torch.equal(v1.batch.last_return, v2.batch.last_return)
print("soak ok")
