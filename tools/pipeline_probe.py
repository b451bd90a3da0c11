#!/usr/bin/env python3
"""Does splitting the batch into S sub-batches on S streams (no per-step global barrier) hide the
per-launch ramp/tail?  Times S in {1,2,4} at the same total drone count, interleaved."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpyv_amd import load_params, sticks
from fpyv_amd.env import DroneBatch

dev = torch.device("cuda:0")
N, ring, launches, rounds = 1 << 20, 32, 400, 5
p = load_params(fps=1000)
acts = sticks.ema_noise_device(ring, N, dev)
cfgs = {}
for S in (1, 2, 4):
    n = N // S
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    envs = [DroneBatch(p, n, device=dev, with_accel=False) for _ in range(S)]
    for e in envs:
        e.reset()
    a = [acts[:, k * n:(k + 1) * n].contiguous() for k in range(S)]
    cfgs[S] = (streams, envs, a)
torch.cuda.synchronize()
times = {S: [] for S in cfgs}
for r in range(rounds + 1):
    for S, (streams, envs, a) in cfgs.items():
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))
        for t in range(launches):
            for k in range(S):
                with torch.cuda.stream(streams[k]):
                    envs[k].step(a[k][t % ring], return_imu=False)
        for st in streams:
            torch.cuda.current_stream(dev).wait_stream(st)
        t1.record(); torch.cuda.synchronize()
        if r:
            times[S].append(t0.elapsed_time(t1) * 1e3 / launches)
for S in cfgs:
    med = statistics.median(times[S])
    print(f"S={S}: {med:7.2f} us per full step  {133 * N / med / 1e3:7.1f} GB/s  {N / med:7.1f} M env-steps/s", flush=True)
