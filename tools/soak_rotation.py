#!/usr/bin/env python3
"""Long-run soak of the rotation of the traversal: two batches of the same seed, one with the automatic rotation and one in the
plain order, stepped by single-step launches (in-kernel noise sticks, auto-reset, episode bookkeeping); every `--check` steps
all their buffers must be bit-identical, finite, with unit quaternions.  The start block wraps round the population thousands
of times on the way; a ragged population exercises the empty blocks of the last round of XCDs.

    python tools/soak_rotation.py [--n 1048576] [--steps 400000] [--check 50000]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpyv_amd import load_params  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=400_000)
ap.add_argument("--check", type=int, default=50_000)
a = ap.parse_args()
p = load_params(fps=1000, ceiling=100.0).replace(noise_gain=1.5)
kw = dict(device="cuda:0", auto_reset=True, stick_noise=True, noise_seed=42, track_episodes=True, with_accel=False)
rot, plain = DroneBatch(p, a.n, **kw), DroneBatch(p, a.n, **kw)
plain.set_rotation(0)
for e in (rot, plain):
    e.reset()
print(f"n = {a.n}, rotation {rot.rotation} drones per launch against the plain order, {a.steps} single-step launches each", flush=True)
t0, done = time.perf_counter(), 0
while done < a.steps:
    k = min(a.check, a.steps - done)
    for e in (rot, plain):
        e.rollout(None, steps=k, fused=False)
    done += k
    torch.cuda.synchronize()
    s = rot.state[:, :a.n]
    same = all(torch.equal(getattr(rot, name)[..., :a.n] if getattr(rot, name).shape[-1] >= a.n else getattr(rot, name), getattr(plain, name)[..., :a.n] if getattr(plain, name).shape[-1] >= a.n else getattr(plain, name))
               for name in ("state", "noise_state", "reward", "done", "ep_return", "ep_length", "last_return", "last_length"))
    qn = torch.linalg.vector_norm(s[6:10], dim=0)
    print(f"{done:8d} steps  {time.perf_counter() - t0:6.1f} s  bit-identical {same}  finite {bool(torch.isfinite(s).all())}  |q|-1 max {float((qn - 1).abs().max()):.2e}  "
          f"episodes ended so far: mean length {float(rot.last_length.float().mean()):.0f}", flush=True)
    assert same and bool(torch.isfinite(s).all()) and float((qn - 1).abs().max()) < 1e-6
print("ok", flush=True)
