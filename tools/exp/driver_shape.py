"""Where the wall clock of the driver's bench shape (`--steps 20 --warmup 5`) goes.

20 launches of the 22.6 us step kernel are 452 us of GPU work; the bench line's wall clock for them is ~500 us.
This probe times the same 20 steps between two device synchronisations in several host shapes, 200 repetitions
each (median / min), so the fixed cost can be attributed: the synchronise itself, the two timing events, the
Python wrapper around each launch, 20 C calls against one.

    python tools/exp/driver_shape.py
"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda", 0)
n, k = 1 << 20, 20
env = DroneBatch(load_params(fps=1000, ceiling=100.0), n, device=dev, auto_reset=True, with_accel=False)
env.reset()
acts = sticks.ema_noise_device(64, n, dev, seed=1234)
for t in range(64):
    env.step(acts[t], return_imu=False)
torch.cuda.synchronize()


def shape_bench(events):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if events:
        ev0.record()
    for t in range(k):
        env.step(acts[t], return_imu=False)
    if events:
        ev1.record()


def shape_raw():
    for t in range(k):
        env._step_raw(acts[t])


def shape_one_call():
    env.rollout(acts[:k], fused=False)


def shape_graph():
    env.rollout(acts[:k], fused=False, graph=True)


def shape_empty():
    pass


def timed(fn, reps=200):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e6)
    return statistics.median(out), min(out)


shape_graph()
torch.cuda.synchronize()
for name, fn in (("synchronise only", shape_empty),
                 ("bench shape: 2 events + 20 env.step", lambda: shape_bench(True)),
                 ("20 env.step, no events", lambda: shape_bench(False)),
                 ("20 _step_raw (one ctypes call each)", shape_raw),
                 ("fpv_rollout: 20 launches from one C call", shape_one_call),
                 ("fpv_rollout_graph: one graph launch of 20 nodes", shape_graph)):
    med, lo = timed(fn)
    print(f"{name:50s} median {med:8.1f} us  min {lo:8.1f} us   -> {n * k / med / 1e3:6.2f} G env-steps/s at the median", flush=True)
