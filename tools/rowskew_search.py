#!/usr/bin/env python3
"""Random search, with the REAL step kernel, over where the 14 state rows sit inside one physically contiguous arena
(variant library built with -DFPV_EXP_ROWSKEW=1: row r at state + r * ld + skew[r]).  Which relative row placements make
HBM fast at 2^23 drones - and is a good one good again in the next process?

    python tools/ab_variants.py --build --only rowskew        # build container
    python tools/rowskew_search.py [configs] [mode] [--replay file]   # GPU box; mode 0 random slots + subs, 1 consecutive slots + subs
"""
import ctypes as C
import json
import os
import random
import statistics
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import torch  # noqa: E402

from fpyv_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(HERE, "_variants", "libfpv_v_rowskew.so")
from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

configs = int(sys.argv[1]) if len(sys.argv) > 1 else 150
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
replay = sys.argv[sys.argv.index("--replay") + 1] if "--replay" in sys.argv else None
dev = torch.device("cuda", 0)
n, R = 1 << 23, 14
MiB = 1 << 20
params = load_params(fps=1000, ceiling=100.0)
env = DroneBatch(params, n, device=dev, auto_reset=True, with_accel=False)
L = env._L
acts = sticks.ema_noise_device(4, n, dev, seed=99)
ld = env.ld
slot_floats, nslots = 33 * MiB // 4, (100 if mode == 3 else 60)
arena = torch.zeros(slot_floats * nslots + MiB, dtype=torch.float32, device=dev)
off = ((arena.data_ptr() + 2 * MiB - 1) // (2 * MiB) * (2 * MiB) - arena.data_ptr()) // 4
env.state = arena[off:off + 14 * ld].view(14, ld)          # only its data_ptr is used: the rows go where the skew says
env._fill_buffers()
print(f"# arena 0x{arena.data_ptr():x} origin +{off * 4} ld {ld} mode {mode}", flush=True)


def set_rows(slots, subs):
    skew = (C.c_int64 * 16)(*[slots[r] * slot_floats + subs[r] - r * ld for r in range(R)], 0, 0)
    assert L.fpv_exp_set_skew(skew) == 0
    env.reset()


def timed(k=12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    env.rollout(acts, fused=False)
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0.record()
        for _ in range(k // 4):
            env.rollout(acts, fused=False)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / k)
    return statistics.median(out)


if mode == 3:
    # consecutive rows at the shipped stride ld plus a NON-UNIFORM extra: row r at r * ld + G[r] * unit (floats)
    n_arg = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else n
    KiB = 1024
    pats = {"uniform": [0] * R, "tri r(r-1)/2": [r * (r - 1) // 2 for r in range(R)], "tri2 r(r+1)/2 mod": [(r * (r + 1) // 2) % 11 + 2 * r for r in range(R)],
            "bitrev4": [int(f"{r:04b}"[::-1], 2) for r in range(R)], "squares r^2 mod 17": [(r * r) % 17 for r in range(R)],
            "fib": [0, 1, 2, 4, 7, 12, 20, 33, 54, 88, 143, 232, 376, 609], "primes": [0, 2, 5, 10, 17, 28, 41, 58, 77, 100, 129, 160, 197, 238],
            "odd gaps 1,3,5..": [r * r for r in range(R)], "rand cum 0-7": None}
    rr = random.Random(5)
    cum, g = [], 0
    for r in range(R):
        cum.append(g); g += rr.randrange(8)
    pats["rand cum 0-7"] = cum
    print(f"# {'pattern':>22s} " + " ".join(f"{u:>9s}" for u in ("64 KiB", "256 KiB", "1 MiB", "2 MiB", "4 MiB")), flush=True)
    for name, G in pats.items():
        row = []
        for unit in (64 * KiB, 256 * KiB, MiB, 2 * MiB, 4 * MiB):
            need = (R - 1) * ld + max(G) * (unit // 4) + ld
            if need > arena.numel() - off:
                row.append(float("nan")); continue
            skew = (C.c_int64 * 16)(*[G[r] * (unit // 4) for r in range(R)], 0, 0)
            assert L.fpv_exp_set_skew(skew) == 0
            env.reset()
            row.append(timed())
        print(f"# {name:>22s} " + " ".join(f"{t:9.2f}" for t in row) + f"   (max G {max(G)})", flush=True)
    sys.exit(0)
rng = random.Random(2024 + mode)
results = []
if replay:
    todo = [(c["slots"], c["subs"]) for c in json.load(open(replay))]
else:
    todo = [(list(range(R)), [r * 256 for r in range(R)])]                    # the shipped layout's analogue: consecutive, 1 KiB apart
    for _ in range(configs):
        slots = rng.sample(range(nslots), R) if mode == 0 else [s + rng.randrange(nslots - R) * 0 for s in range(R)]
        if mode == 4:                            # all 14 rows inside the arena's first 1 GiB (31 slots of 33 MiB), irregular spacing
            slots = rng.sample(range(31), R)
        if mode == 5:                            # rows in the LAST 14 + k slots of the arena
            slots = rng.sample(range(nslots - 20, nslots), R)
        if mode == 1:
            first = rng.randrange(nslots - R)
            slots = [first + r for r in range(R)]
        subs = [rng.randrange(4096) * 64 for _ in range(R)]                  # multiples of 256 B below 1 MiB
        todo.append((slots, subs))
for slots, subs in todo:
    set_rows(slots, subs)
    t = timed()
    results.append({"us": t, "slots": slots, "subs": subs})
    print(f"{t:8.2f} " + " ".join(f"{s}:{b // 64}" for s, b in zip(slots, subs)), flush=True)
ts = sorted(r["us"] for r in results)
print(f"# min {ts[0]:.2f} q1 {ts[len(ts) // 4]:.2f} median {ts[len(ts) // 2]:.2f} q3 {ts[3 * len(ts) // 4]:.2f} max {ts[-1]:.2f}", flush=True)
best = sorted(results, key=lambda r: r["us"])
pick = best[:5] + best[-3:]
print("# again (best five, worst three):", flush=True)
for c in pick:
    set_rows(c["slots"], c["subs"])
    print(f"#   first {c['us']:8.2f}  again {timed():8.2f}", flush=True)
if not replay:
    json.dump(pick, open(os.path.join(os.path.dirname(HERE), "gpurun_out", f"rowskew_pick_m{mode}.json"), "w"))
