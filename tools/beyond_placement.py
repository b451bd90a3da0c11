#!/usr/bin/env python3
"""Where the 2^23-drone state matrix lands, against what a launch costs.  One tool, four experiments:

    python tools/beyond_placement.py combos [K]   K state matrices x K action rings kept alive in one process, every pair timed; then
                                                 freed and allocated again (reverse order, with spacers): whose placement decides?
    python tools/beyond_placement.py sizes [K]    K state allocations per population 2^22 ... 2^24: does the spread shrink beyond the cache?
    python tools/beyond_placement.py reroll [T]   T trials in one process, every second one after empty_cache(): same virtual addresses,
                                                 other physical pages; run it in several processes for the spread between them
    python tools/beyond_placement.py props [K]    K over-sized allocations; inside each: other row strides, the matrix moved by 2 ... 62 MiB

All through the C ABI (fpv_rollout: single-step launches of the shipped kernel), HIP events, medians of three."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpyv_amd import _lib, load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda", 0)
exp = sys.argv[1] if len(sys.argv) > 1 else "combos"
K = int(sys.argv[2]) if len(sys.argv) > 2 else {"combos": 4, "sizes": 5, "reroll": 6, "props": 6}[exp]
params = load_params(fps=1000, ceiling=100.0)
MiB = 1 << 20


def make(n):
    env = DroneBatch(params, n, device=dev, auto_reset=True, with_accel=False)
    return env, sticks.ema_noise_device(4, n, dev, seed=99)


def timed(env, acts, launches=24):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    env.rollout(acts, fused=False)
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0.record()
        for _ in range(launches // 4):
            env.rollout(acts, fused=False)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / launches)
    return statistics.median(out)


def use(env, state, ld=None):
    if ld is not None:
        env.ld = ld
    env.state = state
    env._fill_buffers(); env.reset()


if exp == "combos":
    n = 1 << 23
    env, ring_src = make(n)
    ld = env.ld
    for phase in ("first allocation", "freed, allocated again in reverse order", "freed, allocated again with spacers"):
        if phase != "first allocation":
            del states, actions
            torch.cuda.empty_cache()
        states, actions, keep = [None] * K, [None] * K, []
        if phase.endswith("reverse order"):
            for j in reversed(range(K)):
                actions[j] = ring_src.clone()
            for i in reversed(range(K)):
                states[i] = torch.zeros((14, ld), dtype=torch.float32, device=dev)
        else:
            for i in range(K):
                if phase.endswith("spacers"):
                    keep.append(torch.empty((5 + 13 * i) << 20, dtype=torch.uint8, device=dev))
                states[i] = torch.zeros((14, ld), dtype=torch.float32, device=dev)
                actions[i] = ring_src.clone()
        print(f"--- {phase}: states at " + " ".join(f"0x{s.data_ptr():x}" for s in states) + "; actions at " + " ".join(f"0x{x.data_ptr():x}" for x in actions), flush=True)
        print("            " + "".join(f"  action {j}" for j in range(K)))
        for i in range(K):
            use(env, states[i])
            print(f"  state {i}:  " + "".join(f"  {timed(env, actions[j], 40):8.2f}" for j in range(K)), flush=True)
elif exp == "sizes":
    for n in (1 << 22, 3 << 21, 1 << 23, 3 << 22, 1 << 24):
        env, acts = make(n)
        ld = env.ld
        del env.state
        torch.cuda.empty_cache()
        keep, res = [], []
        for i in range(K):
            keep.append(torch.empty((5 + 13 * i) << 20, dtype=torch.uint8, device=dev))
            st = torch.zeros((14, ld), dtype=torch.float32, device=dev)
            keep.append(st)
            use(env, st)
            res.append(timed(env, acts, 40))
        print(f"n = {n:9d} (state {14 * ld * 4 / 2**20:6.0f} MiB): us " + " ".join(f"{t:7.2f}" for t in res) + "   GB/s " + " ".join(f"{133 * n / t / 1e3:5.0f}" for t in res)
              + f"   spread {100 * (max(res) / min(res) - 1):4.1f} %", flush=True)
        del env, acts, keep, st
        torch.cuda.empty_cache()
elif exp == "reroll":
    n = 1 << 23
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    L = _lib.lib()
    spacers = []
    for trial in range(K):
        if trial % 2 == 1:
            torch.cuda.empty_cache()                      # odd trials: fresh segments from the driver
        if trial >= 2:
            spacers.append(torch.empty((7 + 29 * trial) << 20, dtype=torch.uint8, device=dev))
        env, acts = make(n)
        env.reset()
        step_us = timed(env, acts, 100)
        cf = (133 * n // 8) // 1024 * 1024
        src = torch.empty(cf, dtype=torch.float32, device=dev).normal_(); dst = torch.empty_like(src)
        s = torch.cuda.current_stream().cuda_stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(8):
            _lib.check(L.fpv_diag_stream_copy_wide(dst.data_ptr(), src.data_ptr(), cf, s))
        e0.record()
        for _ in range(40):
            _lib.check(L.fpv_diag_stream_copy_wide(dst.data_ptr(), src.data_ptr(), cf, s))
        e1.record(); torch.cuda.synchronize()
        copy_us = e0.elapsed_time(e1) * 1e3 / 40
        print(f"pid {os.getpid()} trial {trial}: step {step_us:7.2f} us ({133 * n / step_us / 1e3:5.0f} GB/s)  copy16 {copy_us:7.2f} us ({8 * cf / copy_us / 1e3:5.0f} GB/s)  "
              f"ratio {copy_us / step_us:5.3f}  state 0x{env.state.data_ptr():x} action 0x{acts.data_ptr():x} reward 0x{env.reward.data_ptr():x}", flush=True)
        del env, acts, src, dst
elif exp == "props":
    n = 1 << 23
    env, acts = make(n)
    ld0 = env.ld
    cases = [("shipped", 0, ld0)] + [(f"ld n+{p}", 0, n + p) for p in (512, 1024, 65536 + 256, (1 << 20) + 256)] \
        + [(f"shift {m} MiB", m * MiB, ld0) for m in (2, 4, 8, 16, 32, 62)]
    print("allocation            " + "".join(f"{c[0]:>16s}" for c in cases))
    keep = []
    for i in range(K):
        keep.append(torch.empty((5 + 13 * i) << 20, dtype=torch.uint8, device=dev))
        buf = torch.zeros(14 * (n + (1 << 20) + 256) * 4 + 64 * MiB, dtype=torch.uint8, device=dev)
        keep.append(buf)
        row = []
        for name, sh, ld in cases:
            use(env, buf[sh:sh + 14 * ld * 4].view(torch.float32).view(14, ld), ld)
            row.append(timed(env, acts))
        print(f"0x{buf.data_ptr():x} " + "".join(f"{t:16.2f}" for t in row), flush=True)
else:
    raise SystemExit(__doc__)
