"""Row-stride sweep for the fp16-state kernel (and, for comparison, the Kahan kernel) at 2^20 drones: is
fpv_recommended_ld - tuned on the 14 fp32 rows of the plain kernel - also right for 3 fp32 rows + 5 pair rows + 1 half row
in two allocations?

    python tools/exp/ld_sweep_h.py [fp16|kahan|f32]
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from fpyv_amd import _lib, load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda", 0)
geom = sys.argv[1] if len(sys.argv) > 1 else "fp16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
params = load_params(fps=1000, ceiling=100.0)
acts = sticks.ema_noise_device(32, n, dev, seed=3)
base = (-n) % 1024                                   # pads are given relative to the next multiple of 1024 floats (4 KiB)
pads = tuple(base + x for x in (0, 256, 512, 1024, 2048, 4096, 6144, 8192, 10240, 16384))
envs = {}
for pad in pads:
    e = DroneBatch(params, n, device=dev, auto_reset=True, with_accel=False, fp16_state=(geom == "fp16"), kahan_position=(geom == "kahan"))
    ld = n + pad
    e.ld = ld
    f32 = dict(dtype=torch.float32, device=dev)
    if geom == "fp16":
        e.state = torch.zeros((3, ld), **f32)
        e.state_h = torch.zeros(_lib.FPV_HALF_HALVES * ld, dtype=torch.float16, device=dev)
    else:
        e.state = torch.zeros((14, ld), **f32)
        if geom == "kahan":
            e.pos_comp = torch.zeros((6, ld), **f32)
    e._fill_buffers()
    e.reset()
    envs[pad] = e
res = {p: [] for p in pads}
for r in range(7):
    for pad, e in envs.items():
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            e.rollout(acts, fused=False)
        e1.record()
        torch.cuda.synchronize()
        if r:
            res[pad].append(e0.elapsed_time(e1) * 1e3 / (6 * 32))
rec = int(_lib.lib().fpv_recommended_ld(n)) - n
for pad in pads:
    print(f"{geom:6s} ld = n + {pad:5d} floats ({pad * 4:6d} B){' <- fpv_recommended_ld' if pad == rec else '':22s}: median {statistics.median(res[pad]):7.3f} us  min {min(res[pad]):7.3f}", flush=True)
