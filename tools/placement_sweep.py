#!/usr/bin/env python3
"""Where the buffers of the step kernel sit, against what a launch costs, beyond the Infinity Cache (2^23 drones).

Two runs of bench.py on the same box gave 178.7 and 201.1 us per launch for the same kernel at 2^23 drones - 1.01 and 0.90 of
the float4 copy measured beside it (gpurun_out/r5a_bench_20.json / r5a_bench_default.json); the only difference between
the processes is where torch's allocator put the 470 MB state matrix, the action batches and the reward / done rows.  This
tool places every buffer by hand inside ONE arena and times the plain step kernel (through the C ABI, fpv_rollout: single-step
launches) while one placement parameter moves at a time:

  shift   everything moved together by a byte offset (is it the absolute address?)
  ld      the row stride of the state matrix (floats; n + pad)
  action  the action ring's offset relative to a 2 MiB boundary
  reward  the reward row's offset;   done  the done row's offset
  gap     distance between the state matrix and the action ring (MiB)
  stream  the stream the launches go to (null stream, fresh streams, high-priority streams), same buffers

    python tools/placement_sweep.py [--n 8388608] [--sweeps shift ld action reward done gap]
"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 23)
ap.add_argument("--ring", type=int, default=4)
ap.add_argument("--launches", type=int, default=40)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--sweeps", nargs="*", default=["shift", "ld", "action", "reward", "done", "gap"])
a = ap.parse_args()
dev = torch.device("cuda", 0)
n, ring = a.n, a.ring
MiB = 1 << 20
params = load_params(fps=1000, ceiling=100.0)
env = DroneBatch(params, n, device=dev, auto_reset=True, with_accel=False)
ld0 = env.ld
keep = (env.state, env.reward, env.done)           # the batch's own tensors stay alive (and unused)
ring_src = sticks.ema_noise_device(ring, n, dev, seed=99)
BIG_PADS = [256, (1 << 16) + 256, (1 << 18) + 256, (1 << 19) + 256, (1 << 20) + 256, 3 * (1 << 19) + 256, (1 << 21) + 256, 3 * (1 << 20) + 256,
            (1 << 22) + 256, 5 * (1 << 20) + 256, 6 * (1 << 20) + 256, 7 * (1 << 20) + 256, (1 << 23) + 256, 3 * (1 << 22) + 256,
            (1 << 20) + (1 << 16) + 256, (1 << 21) + (1 << 19) + (1 << 17) + 256, 1234567 // 64 * 64 + 256, 2718281 // 64 * 64 + 256, 5555555 // 64 * 64 + 256]
max_pad = max(BIG_PADS) if "ldbig" in a.sweeps else (1 << 18)
arena_bytes = 14 * (n + max_pad) * 4 + ring * n * 16 + n * 5 + 512 * MiB
arena = torch.empty(arena_bytes, dtype=torch.uint8, device=dev)
base = (arena.data_ptr() + 2 * MiB - 1) // (2 * MiB) * (2 * MiB) - arena.data_ptr()       # first 2 MiB boundary inside the arena
print(f"n = {n}, recommended ld = {ld0} (n + {ld0 - n}), arena at 0x{arena.data_ptr():x}, 2 MiB-aligned origin at +{base}", flush=True)


def up(x, m):
    return (x + m - 1) // m * m


def place(shift=0, ld=ld0, a_off=0, r_off=0, d_off=0, gap_mib=0):
    """state | gap | action ring | reward | done, each starting on a 2 MiB boundary (+ its own offset), all moved by `shift`."""
    o = base + shift
    s_bytes = 14 * ld * 4
    st = arena[o:o + s_bytes].view(torch.float32).view(14, ld)
    o2 = up(o - shift + s_bytes, 2 * MiB) + gap_mib * MiB + shift + a_off
    acts = arena[o2:o2 + ring * n * 16].view(torch.float32).view(ring, n, 4)
    o3 = up(o2 - shift - a_off + ring * n * 16, 2 * MiB) + shift + r_off
    rew = arena[o3:o3 + 4 * n].view(torch.float32)
    o4 = up(o3 - shift - r_off + 4 * n, 2 * MiB) + shift + d_off
    done = arena[o4:o4 + n].view(torch.bool)
    assert o4 + n <= arena_bytes
    acts.copy_(ring_src)
    env.ld, env.state, env.reward, env.done, env.done_u8 = ld, st, rew, done, done.view(torch.uint8)
    env._fill_buffers()
    env.reset()
    return acts


def timed(acts):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(1, a.launches // ring)
    env.rollout(acts, fused=False)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        env.rollout(acts, fused=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * ring)


def sweep(name, key, values, fmt):
    res = {v: [] for v in values}
    for r in range(a.rounds):
        for v in values:
            acts = place(**{key: v})
            res[v].append(timed(acts))
    print(f"--- {name}", flush=True)
    for v in values:
        med = statistics.median(res[v])
        print(f"  {key} = {fmt(v):>14s}: median {med:8.2f} us  min {min(res[v]):8.2f}  {133 * n / med / 1e3:7.0f} GB/s", flush=True)


# warm clocks
acts = place()
for _ in range(30):
    env.rollout(acts, fused=False)
torch.cuda.synchronize()
KiB = 1024
if "stream" in a.sweeps:
    # the same buffers, the same kernel, another stream: torch's null stream, fresh streams (each may sit on another hardware
    # queue), high-priority streams
    print("--- the stream the launches go to (same buffers)", flush=True)
    acts = place()
    cand = [("null stream", torch.cuda.default_stream(dev))] + [(f"new stream {k}", torch.cuda.Stream(device=dev)) for k in range(10)] \
        + [(f"high-priority stream {k}", torch.cuda.Stream(device=dev, priority=-1)) for k in range(3)]
    res = {name: [] for name, _ in cand}
    for r in range(a.rounds):
        for name, st in cand:
            with torch.cuda.stream(st):
                res[name].append(timed(acts))
    for name, st in cand:
        med = statistics.median(res[name])
        print(f"  {name:>24s} (0x{st.cuda_stream:x}): median {med:8.2f} us  min {min(res[name]):8.2f}  {133 * n / med / 1e3:7.0f} GB/s", flush=True)
if "shift" in a.sweeps:
    sweep("everything shifted together (bytes)", "shift", [0, 256, 4 * KiB, 64 * KiB, 1 * MiB, 2 * MiB, 34 * MiB, 254 * MiB], lambda v: f"{v}")
if "ld" in a.sweeps:
    pads = [64, 256, 320, 512, 768, 1024, 1536, 2048 + 256, 2048 + 1024, 4096 + 256, 4096 + 2048, 8192 + 256, 16384 + 512, 32768 + 1024, 65536 + 256, 131072 + 512, 131072 + 65536]
    sweep("row stride of the state matrix: ld = n + pad (floats)", "ld", [n + p for p in pads], lambda v: f"n+{v - n}")
if "ldbig" in a.sweeps:
    sweep("row stride, large pads: ld = n + pad (floats)", "ld", [n + p for p in BIG_PADS], lambda v: f"n+{v - n}")
if "action" in a.sweeps:
    sweep("action ring offset from its 2 MiB boundary (bytes)", "a_off", [0, 256, 1 * KiB, 4 * KiB, 16 * KiB, 64 * KiB, 256 * KiB, 1 * MiB], lambda v: f"{v}")
if "reward" in a.sweeps:
    sweep("reward row offset (bytes)", "r_off", [0, 256, 1 * KiB, 4 * KiB, 64 * KiB, 1 * MiB], lambda v: f"{v}")
if "done" in a.sweeps:
    sweep("done row offset (bytes)", "d_off", [0, 256, 1 * KiB, 4 * KiB, 64 * KiB, 1 * MiB], lambda v: f"{v}")
if "gap" in a.sweeps:
    sweep("gap between the state matrix and the action ring (MiB)", "gap_mib", [0, 2, 6, 14, 30, 62, 126, 254], lambda v: f"{v}")
