"""Can the per-launch floor of one half of the population hide behind the streaming of the other half?

Every drone is independent, so a population can be stepped as two half-populations on two streams with no
synchronisation between them: while one half sits in the ~4 us floor between two dependent launches, the other half
could stream.  Variants: (a) one handle, one stream (the product); (b) two handles on two ordinary streams; (c) two
handles on two streams restricted to disjoint halves of the CUs (hipExtStreamCreateWithCUMask: XCDs 0-3 / 4-7), so
that the halves do not compete for wave slots.  Time per step of the WHOLE population, medians over interleaved rounds.

    python tools/exp/two_halves.py
"""
import ctypes as C
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from fpyv_amd import _lib, load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda", 0)
n, k = 1 << 20, 32
params = load_params(fps=1000, ceiling=100.0)
torch.zeros(1, device=dev)                              # loads and initialises the HIP runtime torch ships
_hip_path = next(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln)
hip = C.CDLL(_hip_path)                                 # the SAME runtime instance (already mapped)
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]


def masked_stream(lo_cu, hi_cu):
    words = (C.c_uint32 * 8)()
    for cu in range(lo_cu, hi_cu):
        words[cu // 32] |= 1 << (cu % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s


def plain_stream():
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0          # hipStreamNonBlocking
    return s


def env(m, seed):
    e = DroneBatch(params, m, device=dev, auto_reset=True, with_accel=False)
    e.reset()
    return e, sticks.ema_noise_device(k, m, dev, seed=seed)


L = _lib.lib()
streams = [plain_stream() for _ in range(8)]
masked = {2: [masked_stream(0, 128), masked_stream(128, 256)],
          4: [masked_stream(64 * j, 64 * (j + 1)) for j in range(4)]}
torch_stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def parts(S):
    m = n // S
    return [env(m, 10 * S + j) for j in range(S)]


def launch(e, acts, stream, reps):
    b = e._buf
    b.action = acts.data_ptr()
    b.action_ld = 0
    for _ in range(reps):
        _lib.check(L.fpv_rollout(e._handle, C.byref(b), k, e.n * 4, 0, stream))


def run(envs, strs, reps=40):
    torch.cuda.synchronize()
    for s_ in strs:
        hip.hipStreamSynchronize(s_)
    t0 = time.perf_counter()
    for _ in range(reps):                           # alternate so that every queue always holds work
        for (e, acts), s_ in zip(envs, strs):
            launch(e, acts, s_, 1)
    for s_ in strs:
        hip.hipStreamSynchronize(s_)
    return (time.perf_counter() - t0) * 1e6 / (reps * k)


cases = [("1 handle, torch's stream (the product)", parts(1), [torch_stream]),
         ("1 handle, its own stream", parts(1), streams[:1]),
         ("2 halves, 2 streams", parts(2), streams[:2]),
         ("3 thirds, 3 streams", [env(349440, 30 + j) for j in range(3)], streams[:3]),
         ("4 quarters, 4 streams", parts(4), streams[:4]),
         ("8 eighths, 8 streams", parts(8), streams[:8]),
         ("2 halves, 2 CU-masked streams (128 CUs each)", parts(2), masked[2]),
         ("4 quarters, 4 CU-masked streams (64 CUs each)", parts(4), masked[4])]
res = {c[0]: [] for c in cases}
for r in range(7):
    for name, envs, strs in cases:
        t = run(envs, strs)
        if r:
            res[name].append(t)
for name, envs, _ in cases:
    tot = sum(e.n for e, _ in envs)
    med = statistics.median(res[name])
    print(f"{name:48s}: median {med:7.2f} us per step of {tot} drones   min {min(res[name]):7.2f}   {tot / med / 1e3:6.2f} G env-steps/s", flush=True)
