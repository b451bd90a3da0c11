#!/usr/bin/env python3
"""Workload for rocprofv3 counter passes: L launches of the step kernel at N drones (action ring
beyond the Infinity Cache) + C launches of the known-byte-count calibration copy
(fpv_diag_stream_copy, 2 x 4 x n_floats bytes each).  Run under
    rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -d <dir> -- python3 tools/pmc_probe.py
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpyv_amd import _lib, load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--launches", type=int, default=64)
ap.add_argument("--ring", type=int, default=32)
ap.add_argument("--calib-floats", type=int, default=1 << 27)     # 512 MiB read + 512 MiB written
a = ap.parse_args()
dev = torch.device("cuda:0")
env = DroneBatch(load_params(fps=1000), a.n, device=dev, with_accel=False)
env.reset()
acts = sticks.ema_noise_device(a.ring, a.n, dev)
done = 0
while done < a.launches:
    span = min(a.ring, a.launches - done)
    env.rollout(acts[:span], fused=False)        # single-step launches: the kernel being measured
    done += span
torch.cuda.synchronize()
src = torch.randn(a.calib_floats, device=dev)
dst = torch.empty_like(src)
L = _lib.lib()
for _ in range(4):
    _lib.check(L.fpv_diag_stream_copy(dst.data_ptr(), src.data_ptr(), a.calib_floats, None))
torch.cuda.synchronize()
print("probe done", a.n, a.launches, a.calib_floats)
