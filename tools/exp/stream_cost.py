#!/usr/bin/env python3
"""What do the streams around the state cost in the single-step kernel?  Same env, same launches, interleaved:
reward / done / done_bits outputs switched off one by one, and the action batch served from HBM (ring of 32
distinct 16.8 MB batches, 537 MB > Infinity Cache) or from the cache (one batch re-read every step)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fpyv_amd import load_params, sticks
from fpyv_amd.env import DroneBatch
dev = torch.device("cuda:0"); n = 1 << 20; ring = 32
p = load_params(fps=1000, ceiling=100.0)
acts = sticks.ema_noise_device(ring, n, dev)
env = DroneBatch(p, n, device=dev, with_accel=False, with_done_bits=True, auto_reset=True); env.reset()
ptrs = dict(reward=env._buf.reward, done=env._buf.done, bits=env._buf.done_bits)
cases = {"reward+done (bench)": ("reward", "done"), "reward+done+bits": ("reward", "done", "bits"), "reward+bits": ("reward", "bits"),
         "reward only": ("reward",), "done only": ("done",), "none": (), "reward+done, action from cache (ring 1)": ("reward", "done", "ring1"),
         "none, action from cache": ("ring1",)}
res = {k: [] for k in cases}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(7):
    for k, on in cases.items():
        env._buf.reward = ptrs["reward"] if "reward" in on else None
        env._buf.done = ptrs["done"] if "done" in on else None
        env._buf.done_bits = ptrs["bits"] if "bits" in on else None
        a = acts[:1].expand(ring, n, 4) if "ring1" in on else acts
        torch.cuda.synchronize(); e0.record()
        for rep in range(8):
            if "ring1" in on:
                env.rollout(acts[0], steps=ring, fused=False)      # held action: the same 16.8 MB every step
            else:
                env.rollout(acts, fused=False)
        e1.record(); torch.cuda.synchronize()
        if r: res[k].append(e0.elapsed_time(e1) * 1e3 / (8 * ring))
for k in cases:
    print(f"{k:45s}: {statistics.median(res[k]):.3f} us", flush=True)
