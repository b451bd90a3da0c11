#!/usr/bin/env python3
"""Long-run soak: 10^6 steps of in-kernel noise sticks at 2^20 drones with auto-reset; the state must
stay finite and the quaternions unit (first-order renormalisation must not drift).

    python tools/soak.py [steps] [start_step]

`start_step` sets the handle's 64-bit step counter first - e.g. 4294467296 (= 2^32 - 500 000) makes the run cross the
2^32 boundary at which the 32-bit counter of ABI <= 3 wrapped and the stick-noise stream started to repeat."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpyv_amd import load_params
from fpyv_amd.env import DroneBatch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = 1 << 20
p = load_params(fps=1000, ceiling=100.0).replace(noise_gain=1.5)
env = DroneBatch(p, n, device="cuda:0", auto_reset=True, stick_noise=True, noise_seed=42, track_episodes=True, with_accel=False)
env.reset()
start = int(sys.argv[2]) if len(sys.argv) > 2 else 0
env.set_step_counter(start)
t0 = time.perf_counter(); done = 0
while done < steps:
    k = min(20000, steps - done)
    env.rollout(None, steps=k); done += k
    torch.cuda.synchronize()
    s = env.state[:, :n]
    qn = torch.linalg.vector_norm(s[6:10], dim=0)
    print(f"{done:8d} steps  counter {env.step_counter()} ({'beyond' if env.step_counter() >= 1 << 32 else 'below'} 2^32)  {time.perf_counter() - t0:6.1f} s  finite={bool(torch.isfinite(s).all())}  max||q|-1|={float((qn - 1).abs().max()):.2e}  "
          f"|z|max={float(s[2].abs().max()):.1f}  mean episode length={float(env.last_length.float().mean()):.0f}", flush=True)
    assert bool(torch.isfinite(s).all()) and float((qn - 1).abs().max()) < 1e-6
assert env.step_counter() == start + steps
print("soak ok")
