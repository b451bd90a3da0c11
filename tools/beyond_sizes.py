#!/usr/bin/env python3
"""The placement lottery of the state matrix against the population: K allocations of the state per size, same action ring,
us per launch and GB/s for each.  If the spread came from a part of the state surviving in the 256 MiB Infinity Cache it
would shrink as the state outgrows the cache; if it is DRAM-side (which physical pages back the rows) it stays."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
params = load_params(fps=1000, ceiling=100.0)
for n in (1 << 22, 3 << 21, 1 << 23, 3 << 22, 1 << 24):
    env = DroneBatch(params, n, device=dev, auto_reset=True, with_accel=False)
    acts = sticks.ema_noise_device(4, n, dev, seed=99)
    ld = env.ld
    del env.state
    torch.cuda.empty_cache()
    keep, res = [], []
    for i in range(K):
        keep.append(torch.empty((5 + 13 * i) << 20, dtype=torch.uint8, device=dev))
        st = torch.zeros((14, ld), dtype=torch.float32, device=dev)
        keep.append(st)
        env.state = st
        env._fill_buffers(); env.reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            env.rollout(acts, fused=False)
        torch.cuda.synchronize()
        out = []
        for _ in range(3):
            e0.record()
            for _ in range(10):
                env.rollout(acts, fused=False)
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / 40)
        res.append(statistics.median(out))
    gb = [133 * n / t / 1e3 for t in res]
    print(f"n = {n:9d} (state {14 * ld * 4 / 2**20:6.0f} MiB): us " + " ".join(f"{t:7.2f}" for t in res) + "   GB/s " + " ".join(f"{g:5.0f}" for g in gb)
          + f"   spread {100 * (max(res) / min(res) - 1):4.1f} %", flush=True)
    del env, acts, keep, st
    torch.cuda.empty_cache()
