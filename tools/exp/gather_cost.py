#!/usr/bin/env python3
"""What one done-mask collective costs the stream that launches the step kernels, by mechanism (one GPU, a
communicator of one rank): no collective / torch.distributed all_gather_into_tensor(async_op=True) on RCCL's own
stream / the C ABI's fpv_allgather_done on a side stream ordered with events / a plain device copy on a side
stream (no RCCL at all).  One collective per `block` steps, 2^20 drones, 128 KiB of mask per step."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from fpyv_amd import _lib, load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", RANK="0", WORLD_SIZE="1")
fd1 = os.dup(1)
os.dup2(2, 1)
dist.init_process_group(backend="nccl", device_id=dev)
w = torch.zeros(1, device=dev)
dist.all_reduce(w)
torch.cuda.synchronize()
os.dup2(fd1, 1)
L = _lib.lib()
ident = (C.c_uint8 * _lib.FPV_COMM_ID_BYTES)()
_lib.check(L.fpv_comm_unique_id(ident))
comm = C.c_void_p()
_lib.check(L.fpv_comm_create(ident, 1, 0, 0, C.byref(comm)))

n, block, steps = 1 << 20, int(os.environ.get("BLOCK", "16")), 4096
words = n // 64
env = DroneBatch(load_params(fps=1000, ceiling=100.0), n, device=dev, auto_reset=True, with_accel=False, with_done_bits=True)
env.reset()
acts = sticks.ema_noise_device(32, n, dev, seed=1)
bucket = [torch.zeros((block, words), dtype=torch.int64, device=dev) for _ in range(2)]
out = [torch.zeros((block, words), dtype=torch.int64, device=dev) for _ in range(2)]
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream()


def run(mode):
    pending = [None, None]
    for t in range(steps):
        b, r = divmod(t, block)
        k = b & 1
        if r == 0 and pending[k] is not None:
            if mode == "torch":
                pending[k].wait()
            else:
                main.wait_event(pending[k])
            pending[k] = None
        env.set_done_bits_target(bucket[k][r].data_ptr())
        env.step(acts[t % 32], return_imu=False)
        if r == block - 1 and mode != "none":
            if mode == "torch":
                pending[k] = dist.all_gather_into_tensor(out[k].view(-1), bucket[k].view(-1), async_op=True)
            else:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                if mode == "abi":
                    _lib.check(L.fpv_allgather_done(comm, bucket[k].data_ptr(), out[k].data_ptr(), block * words, side.cuda_stream))
                else:
                    with torch.cuda.stream(side):
                        out[k].copy_(bucket[k], non_blocking=True)
                done = torch.cuda.Event()
                done.record(side)
                pending[k] = done


# ---- finer: which part of the cross-stream ordering costs?  raw HIP events through ctypes
hip = C.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
DISABLE_TIMING, NO_SYSTEM_FENCE = 0x2, 0x20000000


def raw_event(flags):
    e = C.c_void_p()
    assert hip.hipEventCreateWithFlags(C.byref(e), flags) == 0
    return e


def run_raw(flags, with_side_work, with_wait_back):
    evs = [[raw_event(flags), raw_event(flags)] for _ in range(2)]
    pend = [False, False]
    ms, ss = C.c_void_p(main.cuda_stream), C.c_void_p(side.cuda_stream)
    for t in range(steps):
        b, r = divmod(t, block)
        k = b & 1
        if r == 0 and pend[k]:
            if with_wait_back:
                assert hip.hipStreamWaitEvent(ms, evs[k][1], 0) == 0
            pend[k] = False
        env.set_done_bits_target(bucket[k][r].data_ptr())
        env.step(acts[t % 32], return_imu=False)
        if r == block - 1:
            assert hip.hipEventRecord(evs[k][0], ms) == 0
            if with_side_work:
                assert hip.hipStreamWaitEvent(ss, evs[k][0], 0) == 0
                _lib.check(L.fpv_allgather_done(comm, bucket[k].data_ptr(), out[k].data_ptr(), block * words, side.cuda_stream))
                assert hip.hipEventRecord(evs[k][1], ss) == 0
                pend[k] = True


for name, fl, sw, wb in (("record only, default event", DISABLE_TIMING, False, False),
                         ("record only, no system fence", DISABLE_TIMING | NO_SYSTEM_FENCE, False, False),
                         ("record + side collective, default events", DISABLE_TIMING, True, True),
                         ("record + side collective, no system fence", DISABLE_TIMING | NO_SYSTEM_FENCE, True, True),
                         ("record + side collective, no fence, main never waits back", DISABLE_TIMING | NO_SYSTEM_FENCE, True, False)):
    run_raw(fl, sw, wb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run_raw(fl, sw, wb)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    print(f"block {block:4d}  {name:58s}: {us:7.3f} us per step -> {(us - 22.9) * block:6.1f} us per collective", flush=True)

for mode in ("none", "torch", "abi", "copy", "none"):
    run(mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(mode)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    print(f"block {block:4d}  {mode:6s}: {us:7.3f} us per step   -> {(us - 22.9) * block:6.1f} us per collective over a 22.9 us step", flush=True)
L.fpv_comm_destroy(comm)
dist.destroy_process_group()
