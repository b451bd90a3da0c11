#!/usr/bin/env python3
"""Which buffer's placement decides the 2^23-drone launch time?  K state matrices and K action rings are allocated once
(each its own driver allocation), every (state_i, action_j) pair is timed with the same reward / done rows; then everything
is freed, allocated again in reverse order and timed again (the driver hands the physical pages out in another order)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda", 0)
n, K = 1 << 23, int(sys.argv[1]) if len(sys.argv) > 1 else 4
params = load_params(fps=1000, ceiling=100.0)
env = DroneBatch(params, n, device=dev, auto_reset=True, with_accel=False)
ld = env.ld
ring_src = sticks.ema_noise_device(4, n, dev, seed=99)


def timed(acts, k=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    env.rollout(acts, fused=False)
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0.record()
        for _ in range(k // 4):
            env.rollout(acts, fused=False)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / k)
    return statistics.median(out)


def use(state):
    env.state = state
    env._fill_buffers()
    env.reset()


for phase in ("first allocation", "freed, allocated again in reverse order", "freed, allocated again with spacers"):
    if phase != "first allocation":
        del states, actions
        torch.cuda.empty_cache()
    order = list(range(K))
    states, actions = [None] * K, [None] * K
    keep = []
    if phase.endswith("reverse order"):
        for j in reversed(order):
            actions[j] = ring_src.clone()
        for i in reversed(order):
            states[i] = torch.zeros((14, ld), dtype=torch.float32, device=dev)
    else:
        for i in order:
            if phase.endswith("spacers"):
                keep.append(torch.empty((5 + 13 * i) << 20, dtype=torch.uint8, device=dev))
            states[i] = torch.zeros((14, ld), dtype=torch.float32, device=dev)
            actions[i] = ring_src.clone()
    print(f"--- {phase}: states at " + " ".join(f"0x{s.data_ptr():x}" for s in states) + "; actions at " + " ".join(f"0x{x.data_ptr():x}" for x in actions), flush=True)
    print("            " + "".join(f"  action {j}" for j in range(K)))
    for i in range(K):
        use(states[i])
        print(f"  state {i}:  " + "".join(f"  {timed(actions[j]):8.2f}" for j in range(K)), flush=True)
