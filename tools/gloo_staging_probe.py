#!/usr/bin/env python3
"""What the gloo FALLBACK of the done-mask exchange costs per bucket, two ranks on one GPU: torch's own gloo path for device
tensors against explicit staging through pinned host memory (copy out, gloo on CPU tensors, copy back).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 tools/gloo_staging_probe.py
"""
import time

import torch
import torch.distributed as dist

dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for mib in (2, 8):
    n = mib << 17                                                  # int64 words
    x = torch.zeros(n, dtype=torch.int64, device=dev)
    out = torch.zeros(w * n, dtype=torch.int64, device=dev)
    hx = torch.zeros(n, dtype=torch.int64).pin_memory()
    hout = torch.zeros(w * n, dtype=torch.int64).pin_memory()
    side = torch.cuda.Stream(device=dev)

    def direct():
        dist.all_gather_into_tensor(out, x, async_op=True).wait()

    def staged():
        ev = torch.cuda.Event()
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            hx.copy_(x, non_blocking=True)
            ev.record()
        ev.synchronize()
        dist.all_gather_into_tensor(hout, hx, async_op=True).wait()
        with torch.cuda.stream(side):
            out.copy_(hout, non_blocking=True)
        torch.cuda.current_stream(dev).wait_stream(side)

    for name, fn in (("torch gloo on device tensors", direct), ("staged through pinned host memory", staged)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8
        if r == 0:
            print(f"{mib} MiB per rank, {name}: {dt * 1e3:.2f} ms per gather", flush=True)
dist.destroy_process_group()
