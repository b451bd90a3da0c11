"""Synthetic stick (action) profiles for tests and benchmarks.

Action layout is the reference's: [roll, pitch, yaw, throttle], each nominally in [-1, 1]
(/root/reference/src/utils/components.py:181-186).  Host-side NumPy generators produce float64
and round through float32 so the fp32 device path and the float64 oracle see identical inputs.

* `ema_noise`: the profile of /root/reference/tests/noise_smooth_test.py:6-12 -
  x ~ N(0,1), x_s <- 0.9*x_s + 0.1*x, x_s(0) = 0 - per drone and channel.
* `sinusoid`: constant throttle + sin/cos roll/pitch (BASELINE config 2).
* `ema_noise_device`: the same EMA profile generated with torch on the device, for batches that
  are too large to come from the host (BASELINE config 3).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np


def _f32(a: np.ndarray) -> np.ndarray:
    return a.astype(np.float32)


def zeros(steps: int, n: int) -> np.ndarray:
    return np.zeros((steps, n, 4), dtype=np.float32)


def constant(steps: int, n: int, action: Sequence[float]) -> np.ndarray:
    return np.broadcast_to(_f32(np.asarray(action, dtype=np.float64)), (steps, n, 4)).copy()


def sinusoid(steps: int, n_total: int, dt: float, amplitude: float = 0.3, freq_hz: float = 1.0,
             throttle: float = -0.6, drone_ids: Optional[Sequence[int]] = None) -> np.ndarray:
    """a_i(t) = [A sin(2 pi f t + phi_i), A cos(2 pi f t + phi_i), 0, throttle], phi_i = 2 pi i / n_total,
    t = step * dt.  `drone_ids` selects a subset of the n_total drones (global ids)."""
    ids = np.arange(n_total) if drone_ids is None else np.asarray(drone_ids)
    t = np.arange(steps, dtype=np.float64)[:, None] * dt
    ph = 2 * math.pi * freq_hz * t + 2 * math.pi * ids[None, :] / n_total
    a = np.zeros((steps, len(ids), 4))
    a[..., 0] = amplitude * np.sin(ph)
    a[..., 1] = amplitude * np.cos(ph)
    a[..., 3] = throttle
    return _f32(a)


def ema_noise(steps: int, drone_ids: Sequence[int], seed: int = 0, transition: float = 0.1,
              clip: Optional[float] = None) -> np.ndarray:
    """EMA-smoothed Gaussian sticks.  Drone `i` draws from default_rng(seed + i), so its stick
    history does not depend on how many other drones exist or which rank owns it."""
    ids = list(drone_ids)
    x = np.empty((steps, len(ids), 4))
    for k, i in enumerate(ids):
        x[:, k] = np.random.default_rng(seed + int(i)).standard_normal((steps, 4))
    out = np.empty_like(x)
    s = np.zeros((len(ids), 4))
    for t in range(steps):
        s = s * (1 - transition) + x[t] * transition
        out[t] = s
    if clip is not None:
        out = np.clip(out, -clip, clip)
    return _f32(out)


def ema_noise_device(steps: int, n: int, device, seed: int = 1234, transition: float = 0.1,
                     clip: float = 1.0):
    """[steps, n, 4] float32 torch tensor on `device`, same recurrence as `ema_noise`, one
    torch.Generator stream for the whole batch (used for throughput runs, not for parity)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((steps, n, 4), dtype=torch.float32, device=device)
    s = torch.zeros((n, 4), dtype=torch.float32, device=device)
    for t in range(steps):
        x = torch.randn((n, 4), generator=g, dtype=torch.float32, device=device)
        s = s * (1 - transition) + x * transition
        out[t] = s.clamp(-clip, clip)
    return out
