#!/usr/bin/env python3
"""The 2^23-drone step kernel on a state matrix whose PHYSICAL backing is chosen by hand (HIP virtual memory management,
tools/vmm_alloc.hip): one virtually contiguous [14][ld] matrix mapped onto physical chunks of 2 MiB ... 512 MiB, created in
address order, in reverse order or in a pseudo-random order.  Same virtual layout, same kernel, same action ring."""
import ctypes as C
import os
import statistics
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import torch  # noqa: E402

from fpyv_amd import load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 23
params = load_params(fps=1000, ceiling=100.0)
env = DroneBatch(params, n, device=dev, auto_reset=True, with_accel=False)
acts = sticks.ema_noise_device(4, n, dev, seed=99)
V = C.CDLL(os.path.join(HERE, "_variants", "libvmm.so"))
V.vmm_alloc.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_uint, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
V.vmm_granularity.restype = C.c_size_t
print("granularity", V.vmm_granularity(0), flush=True)
MiB = 1 << 20
total = 14 * env.ld * 4


def timed():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    env.rollout(acts, fused=False)
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0.record()
        for _ in range(6):
            env.rollout(acts, fused=False)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / 24)
    return statistics.median(out)


base_t = timed()
print(f"torch allocation (as shipped): {base_t:8.2f} us  {133 * n / base_t / 1e3:6.0f} GB/s", flush=True)
row_bytes = env.ld * 4
KiB = 1024
chunks = (("2 MiB", 2 * MiB), ("8 MiB", 8 * MiB), ("32 MiB", 32 * MiB), ("one row", row_bytes), ("64 MiB", 64 * MiB), ("128 MiB", 128 * MiB), ("256 MiB", 256 * MiB), ("whole", total))
orders = ((0, "in order"), (1, "reverse"), (2, "random a"), (2, "random b"), (0, "in order + spacers"))
if len(sys.argv) > 2 and sys.argv[2] == "small":
    chunks = (("64 KiB", 64 * KiB), ("256 KiB", 256 * KiB), ("1 MiB", MiB), ("2 MiB", 2 * MiB), ("4 MiB", 4 * MiB), ("8 MiB", 8 * MiB), ("16 MiB", 16 * MiB), ("whole", total))
    orders = ((0, "in order"), (2, "random a"), (0, "in order + spacers"))
if len(sys.argv) > 2 and sys.argv[2] == "brief":
    # one line per process: the shipped allocation against 2 / 4 / 8 MiB chunks and one whole chunk - run it in many processes
    row = [f"torch {base_t:7.2f}"]
    for name, chunk, order in (("2MiB", 2 * MiB, 0), ("2MiB-rnd", 2 * MiB, 2), ("4MiB", 4 * MiB, 0), ("8MiB", 8 * MiB, 0), ("whole", total, 0), ("2MiB-again", 2 * MiB, 0)):
        va, hd = C.c_void_p(), C.c_void_p()
        assert V.vmm_alloc(0, total, chunk, order, 11, 0, C.byref(va), C.byref(hd)) == 0
        env._fill_buffers(); env._buf.state = va.value; env.reset()
        row.append(f"{name} {timed():7.2f}")
        torch.cuda.synchronize(); V.vmm_free(hd)
    print(f"pid {os.getpid()}: " + "  ".join(row), flush=True)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "spacers":
    # one physical chunk per ROW, a spacer allocation of X MiB made before each chunk and held until all rows exist (then freed):
    # does it take a different physical REGION per row to be fast, and how far apart?
    for sp in (0, 32, 128, 512, 1024, 2048, 4096):
        ts = []
        for rep in range(3):
            va, hd = C.c_void_p(), C.c_void_p()
            assert V.vmm_alloc(0, total, row_bytes, 0, 0, sp, C.byref(va), C.byref(hd)) == 0
            env._fill_buffers(); env._buf.state = va.value; env.reset()
            ts.append(timed())
            torch.cuda.synchronize(); V.vmm_free(hd)
        print(f"one chunk per row, spacer {sp:5d} MiB: " + " ".join(f"{t:8.2f}" for t in ts) + " us", flush=True)
    for rows_per_chunk in (2, 4, 7):
        ts = []
        for rep in range(2):
            va, hd = C.c_void_p(), C.c_void_p()
            assert V.vmm_alloc(0, total, rows_per_chunk * row_bytes, 0, 0, 1024, C.byref(va), C.byref(hd)) == 0
            env._fill_buffers(); env._buf.state = va.value; env.reset()
            ts.append(timed())
            torch.cuda.synchronize(); V.vmm_free(hd)
        print(f"{rows_per_chunk} rows per chunk, spacer 1024 MiB: " + " ".join(f"{t:8.2f}" for t in ts) + " us", flush=True)
    sys.exit(0)
for chunk_name, chunk in chunks:
    for order, oname in orders:
        va, hd = C.c_void_p(), C.c_void_p()
        rc = V.vmm_alloc(0, total, chunk, order, 7 if oname.endswith("a") else 99, 3 if "spacers" in oname else 0, C.byref(va), C.byref(hd))
        if rc != 0:
            print(chunk_name, oname, "vmm_alloc failed"); continue
        env._fill_buffers()
        env._buf.state = va.value
        env.reset()
        t = timed()
        print(f"chunk {chunk_name:>8s} {oname:>18s}: {t:8.2f} us  {133 * n / t / 1e3:6.0f} GB/s", flush=True)
        torch.cuda.synchronize()
        V.vmm_free(hd)
