"""Is the 22.2 - 22.9 us spread of the headline launch between processes a property of WHERE the state landed?

Eight DroneBatch objects of 2^20 drones are kept alive in one process (so each state matrix sits at another address, with
other physical pages behind it) and timed interleaved on the same action ring.  If the spread inside one process is as
wide as the spread between processes, placement explains it.

    python tools/exp/placement_reroll.py
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda", 0)
n = 1 << 20
params = load_params(fps=1000, ceiling=100.0)
acts = sticks.ema_noise_device(32, n, dev, seed=3)
envs, pads = [], []
for j in range(8):
    e = DroneBatch(params, n, device=dev, auto_reset=True, with_accel=False, fp16_state=(len(sys.argv) > 1 and sys.argv[1] == "fp16"))
    e.reset()
    envs.append(e)
    pads.append(torch.empty((3 + 5 * j) << 18, dtype=torch.float32, device=dev))      # odd-sized spacers: the next state lands elsewhere
res = [[] for _ in envs]
for r in range(9):
    for j, e in enumerate(envs):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            e.rollout(acts, fused=False)
        e1.record()
        torch.cuda.synchronize()
        if r:
            res[j].append(e0.elapsed_time(e1) * 1e3 / (8 * 32))
for j, e in enumerate(envs):
    extra = f"  state_h - state = {(e.state_h.data_ptr() - e.state.data_ptr()) / 2 ** 20:8.1f} MiB" if e.state_h is not None else ""
    print(f"state at 0x{e.state.data_ptr():x} (+{(e.state.data_ptr() - envs[0].state.data_ptr()) / 2 ** 20:9.1f} MiB): "
          f"median {statistics.median(res[j]):7.3f} us  min {min(res[j]):7.3f}" + extra, flush=True)
meds = [statistics.median(v) for v in res]
print(f"spread inside this process: {min(meds):.3f} .. {max(meds):.3f} us ({100 * (max(meds) / min(meds) - 1):.1f} %)")
