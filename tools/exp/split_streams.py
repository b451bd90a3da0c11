#!/usr/bin/env python3
"""Which streams should the partitions of FpvVecEnv(partitions=2) step on?  The two chains only overlap when their
streams sit on different hardware queues, and the runtime hands hardware queues to streams in its own way.  For each way
of making the streams, several fresh instances in one process, each timed for the same open-loop run (2^20 drones,
a ring of sticks, wall clock between synchronises, median of 3 repetitions).

    python tools/exp/split_streams.py          (run again with GPU_MAX_HW_QUEUES=8 in the environment to compare)
"""
import ctypes as C
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd import env as fenv  # noqa: E402
from fpyv_amd.streams import chain_time_ratio, overlapping_streams  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
n, K = 1 << 20, 1000
params = load_params(fps=1000, ceiling=100.0)
ring = sticks.ema_noise_device(32, n, dev, seed=1234)
hip = C.CDLL("libamdhip64.so")


def raw_stream(flags=1, prio=None):
    s = C.c_void_p()
    if prio is None:
        rc = hip.hipStreamCreateWithFlags(C.byref(s), C.c_uint(flags))
    else:
        rc = hip.hipStreamCreateWithPriority(C.byref(s), C.c_uint(flags), C.c_int(prio))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


_picked = {}


def picked(k):
    if k == 0:
        _picked["s"], _picked["rep"] = overlapping_streams(dev, 2, avoid=[torch.cuda.current_stream(dev)])
        print("   overlapping_streams:", _picked["rep"], flush=True)
    return _picked["s"][k]


MAKERS = {
    "fpyv_amd.streams.overlapping_streams (measured)": picked,
    "torch.cuda.Stream()": lambda k: torch.cuda.Stream(device=dev),
    "torch.cuda.Stream(priority=-1)": lambda k: torch.cuda.Stream(device=dev, priority=-1),
    "hipStreamCreateWithFlags(nonblocking)": lambda k: raw_stream(1),
    "hipStreamCreateWithPriority(nonblocking, -1)": lambda k: raw_stream(1, -1),
    "partition 0 on the caller's stream, partition 1 on a new torch stream": lambda k: torch.cuda.current_stream(dev) if k == 0 else torch.cuda.Stream(device=dev),
}


def run(parts, maker):
    env = fenv.FpvVecEnv(params, num_envs=n, device=dev, track_episodes=False, partitions=parts)
    if parts > 1 and maker is not None:
        for k, P in enumerate(env._parts):
            P.stream = maker(k)
            P._stream_ptr = P.stream.cuda_stream
    probe = None
    if parts > 1:
        probe = (chain_time_ratio(env._parts[0].stream, env._parts[1].stream),
                 max(chain_time_ratio(P.stream, torch.cuda.current_stream(dev)) for P in env._parts if P.stream != torch.cuda.current_stream(dev)))
    env.reset()
    if parts == 1:
        def fn(k):
            for t in range(k):
                env.batch._step_raw(ring[t % 32])
    else:
        rng = [env.partition_range(p) for p in range(env.partitions)]
        sl = [[ring[r][lo:hi] for lo, hi in rng] for r in range(32)]

        def fn(k):
            for t in range(k):
                row = sl[t % 32]
                for p in range(env.partitions):
                    env.step_async(p, row[p], ready=True)
            for p in range(env.partitions):
                env.step_wait(p)
    fn(100)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn(K)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e6)
    env.close()
    if probe is not None:
        return f"{statistics.median(ts):6.2f} (probe {probe[0]:.2f}, vs caller {probe[1]:.2f})"
    return f"{statistics.median(ts):6.2f}"


print(f"# tools/exp/split_streams.py  GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}  (us per step of {n} drones, one number per fresh instance)")
print(f"{'1 partition (env.step)':<75s}: " + "  ".join(run(1, None) for _ in range(4)), flush=True)
for name, mk in MAKERS.items():
    print(f"{'2 partitions, ' + name:<75s}: " + "  ".join(run(2, mk) for _ in range(6)), flush=True)
print(f"{'1 partition (env.step)':<75s}: " + "  ".join(run(1, None) for _ in range(4)), flush=True)
