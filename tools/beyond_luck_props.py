#!/usr/bin/env python3
"""Is the fast / slow placement of the 2^23-drone state matrix a property of the ALLOCATION (every layout inside it is fast,
or slow) or of how the 14 rows sit inside it?  K allocations of 14 rows + 64 MiB of slack; inside each: the shipped layout,
other row strides, the whole matrix moved by 2 ... 62 MiB."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = 1 << 23
params = load_params(fps=1000, ceiling=100.0)
env = DroneBatch(params, n, device=dev, auto_reset=True, with_accel=False)
acts = sticks.ema_noise_device(4, n, dev, seed=99)
ld0 = env.ld
MiB = 1 << 20


def timed():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    env.rollout(acts, fused=False)
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0.record()
        for _ in range(6):
            env.rollout(acts, fused=False)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / 24)
    return statistics.median(out)


def use(buf, shift_bytes, ld):
    env.ld = ld
    env.state = buf[shift_bytes:shift_bytes + 14 * ld * 4].view(torch.float32).view(14, ld)
    env._fill_buffers(); env.reset()


keep = []
cases = [("shipped", 0, ld0)] + [(f"ld n+{p}", 0, n + p) for p in (512, 1024, 65536 + 256, (1 << 20) + 256)] \
    + [(f"shift {m} MiB", m * MiB, ld0) for m in (2, 4, 8, 16, 32, 62)]
print("allocation            " + "".join(f"{c[0]:>16s}" for c in cases))
for i in range(K):
    keep.append(torch.empty((5 + 13 * i) << 20, dtype=torch.uint8, device=dev))
    buf = torch.zeros(14 * (n + (1 << 20) + 256) * 4 + 64 * MiB, dtype=torch.uint8, device=dev)
    keep.append(buf)
    row = []
    for name, sh, ld in cases:
        use(buf, sh, ld)
        row.append(timed())
    print(f"0x{buf.data_ptr():x} " + "".join(f"{t:16.2f}" for t in row), flush=True)
