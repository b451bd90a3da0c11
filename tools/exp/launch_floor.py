"""What one launch of the step kernel costs as a function of the population: t(n) = t0 + n / rate.

Every step kernel depends on the previous one (same state, same stream), so a chain of launches pays the GPU's
launch-to-launch floor - dispatch, the first waves' load latency before any store traffic exists, the last waves'
stores, the end-of-kernel release - once per step, whatever the population.  This probe times chains of single-step
launches (issued from ONE C call, fpv_rollout, so the host is not in the picture) for populations from 4096 drones
to 2^21 (state still inside the 256 MiB Infinity Cache) and fits the line.

    python tools/exp/launch_floor.py
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda", 0)
params = load_params(fps=1000, ceiling=100.0)
geom = sys.argv[1] if len(sys.argv) > 1 else "f32"        # f32 | fp16 | kahan | racerW | racerD
rows = []
sizes = (1 << 12, 1 << 14, 1 << 16, 1 << 17, 1 << 18, 5 << 16, 6 << 16, 7 << 16, 1 << 19, 3 << 18, 1 << 20, 5 << 18, 3 << 19, 7 << 18, 1 << 21)
if geom != "f32":
    sizes = (1 << 12, 1 << 18, 1 << 19, 3 << 18, 1 << 20, 3 << 19, 1 << 21)


def make(n):
    if geom.startswith("racer"):
        from fpyv_amd.env import RacerBatch
        rp = params.replace(mode=1, racer_pid=np.asarray([[0.004, 0.02, 1e-6], [0.003, 0.01, 2e-6], [0.002, 0.005, 0.0]]),
                            racer_omega_dt=(geom == "racerD"), ceiling=100.0)
        return RacerBatch(rp, n, device=dev, auto_reset=True)
    return DroneBatch(params, n, device=dev, auto_reset=True, with_accel=False, fp16_state=(geom == "fp16"),
                      kahan_position=(geom == "kahan"))


print(f"# kernel family: {geom}")
for n in sizes:
    env = make(n)
    env.reset()
    k = 32
    acts = sticks.ema_noise_device(k, n, dev, seed=7)
    for _ in range(4):
        env.rollout(acts, fused=False)
    torch.cuda.synchronize()
    per = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            env.rollout(acts, fused=False)           # 8 x 32 dependent single-step launches
        e1.record()
        torch.cuda.synchronize()
        per.append(e0.elapsed_time(e1) * 1e3 / (8 * k))
    us = statistics.median(per)
    mb = env.algorithmic_bytes() * n / 1e6
    rows.append((n, us, mb))
    print(f"n = {n:8d}  {us:7.3f} us per launch   {mb:8.2f} MB algorithmic   {mb / us:6.3f} TB/s   {n // 64:6d} waves", flush=True)
    del env, acts

big = [(n, us, mb) for n, us, mb in rows if n >= 1 << 19]
A = np.array([[1.0, mb] for _, _, mb in big])
y = np.array([us for _, us, _ in big])
(t0, slope), res, *_ = np.linalg.lstsq(A, y, rcond=None)
print(f"fit over n >= 2^19:  t(n) = {t0:.2f} us + bytes / {1.0 / slope:.2f} TB/s    (max residual {np.abs(A @ [t0, slope] - y).max():.2f} us)")
n20 = next(r for r in rows if r[0] == 1 << 20)
print(f"at 2^20 drones: {n20[1]:.2f} us = {t0:.2f} us floor ({100 * t0 / n20[1]:.0f} %) + {n20[1] - t0:.2f} us of streaming at "
      f"{n20[2] / (n20[1] - t0):.2f} TB/s")
print(f"smallest population: {rows[0][1]:.2f} us per dependent launch at {rows[0][0]} drones (the floor itself)")
