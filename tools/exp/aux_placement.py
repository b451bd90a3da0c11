"""Does it matter WHERE the auxiliary row buffers (Kahan rows, noise state) live relative to the state matrix?

The state rows are padded (fpv_recommended_ld) so that consecutive rows of ONE matrix do not fall on the same channels.
A separately allocated [6, ld] or [4, ld] buffer starts on its own 2 MiB boundary: its row k then has the same offset
pattern as state row k.  This probe times the Kahan and the in-kernel-noise single-step kernels with the auxiliary rows
(a) allocated separately, as DroneBatch does, and (b) carved out of one allocation right behind the state rows (rows
14.. of the same padded matrix), in one process, interleaved.

    python tools/exp/aux_placement.py
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


def make(kind, contiguous, shift_rows=0):
    kw = dict(device=dev, auto_reset=True, with_accel=False)
    if kind == "kahan":
        kw["kahan_position"] = True
    else:
        kw["stick_noise"] = True
    e = DroneBatch(params, n, **kw)
    if contiguous:
        aux_rows = 6 if kind == "kahan" else 4
        big = torch.zeros((14 + shift_rows + aux_rows, e.ld), dtype=torch.float32, device=dev)
        e.state = big[:14]
        aux = big[14 + shift_rows:]
        if kind == "kahan":
            e.pos_comp = aux
        else:
            e.noise_state = aux
        e._fill_buffers()
        e._keep_big = big
    e.reset()
    return e


def timed(e, reps=6):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        e.rollout(acts, fused=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * acts.shape[0])


for kind in ("kahan", "noise"):
    envs = {"separate (as allocated today)": make(kind, False), "behind the state rows": make(kind, True),
            "behind the state rows + 1 spare row": make(kind, True, 1), "a second separate set": make(kind, False)}
    res = {k: [] for k in envs}
    for r in range(9):
        for k, e in envs.items():
            t = timed(e)
            if r:
                res[k].append(t)
    for k, v in res.items():
        e = envs[k]
        aux = e.pos_comp if kind == "kahan" else e.noise_state
        d = (aux.data_ptr() - e.state.data_ptr())
        print(f"{kind:6s} {k:38s}: median {statistics.median(v):7.3f} us  min {min(v):7.3f}   aux - state = {d / 2 ** 20:10.3f} MiB "
              f"({d % (2 << 20)} B past a 2 MiB boundary)", flush=True)
    del envs
