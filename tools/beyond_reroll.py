#!/usr/bin/env python3
"""Is the 178 ... 201 us spread of the 2^23-drone launch between PROCESSES reproducible inside one process by allocating the
batch again (fresh hipMalloc segments after empty_cache, or cached blocks)?  Prints one line per trial; run it several times
in one gpurun call to see the spread between processes next to the spread inside each."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpyv_amd import _lib, load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda", 0)
n = 1 << 23
params = load_params(fps=1000, ceiling=100.0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
L = _lib.lib()


def timed(fn, k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(8)
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0.record(); fn(k); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / k)
    return statistics.median(out)


spacers = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    if trial % 2 == 1:
        torch.cuda.empty_cache()                      # odd trials: fresh segments from the driver
    if trial >= 2:
        spacers.append(torch.empty((7 + 29 * trial) << 20, dtype=torch.uint8, device=dev))
    env = DroneBatch(params, n, device=dev, auto_reset=True, with_accel=False)
    env.reset()
    acts = sticks.ema_noise_device(4, n, dev, seed=99)
    step_us = timed(lambda k: [env.rollout(acts, fused=False) for _ in range(k // 4)], 100)
    cf = (133 * n // 8) // 1024 * 1024
    src = torch.empty(cf, dtype=torch.float32, device=dev).normal_(); dst = torch.empty_like(src)
    s = torch.cuda.current_stream().cuda_stream
    copy_us = timed(lambda k: [_lib.check(L.fpv_diag_stream_copy_wide(dst.data_ptr(), src.data_ptr(), cf, s)) for _ in range(k)], 40)
    print(f"pid {os.getpid()} trial {trial}: step {step_us:7.2f} us ({133 * n / step_us / 1e3:5.0f} GB/s)  copy16 {copy_us:7.2f} us ({8 * cf / copy_us / 1e3:5.0f} GB/s)  "
          f"ratio {copy_us / step_us:5.3f}  state 0x{env.state.data_ptr():x} action 0x{acts.data_ptr():x} reward 0x{env.reward.data_ptr():x}", flush=True)
    del env, acts, src, dst
