import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fpyv_amd import load_params, sticks
from fpyv_amd.env import DroneBatch
dev = torch.device("cuda:0")
n = 1 << 20
env = DroneBatch(load_params(fps=1000, ceiling=100.0), n, device=dev, auto_reset=True, with_accel=False)
env.reset()
acts = sticks.ema_noise_device(32, n, dev, seed=1)
for mode in ("stream", "held", "stream", "held"):
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for j in range(40):
            if mode == "stream": env.rollout(acts)
            else: env.rollout(acts[j % 32], steps=32)
        e1.record(); torch.cuda.synchronize()
    print(mode, f"{e0.elapsed_time(e1) * 1e3 / (40 * 32):.3f} us per env-step", flush=True)
