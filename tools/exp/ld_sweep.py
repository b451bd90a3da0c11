#!/usr/bin/env python3
"""Row stride (ld) of the SoA state vs step-kernel time, beyond the Infinity Cache (2^23 drones) and inside it (2^20):
does the padding rule of fpv_recommended_ld (keep the stride >= 1 KiB past a multiple of 8 KiB) still hold when all
28 row streams run from HBM?"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fpyv_amd import _lib, load_params, sticks
dev = torch.device("cuda:0"); torch.zeros(1, device=dev)
L = _lib.lib()
p = load_params(fps=1000, ceiling=100.0); cp = _lib.pack_params(p, auto_reset=True)
for n in (1 << 23, 1 << 20):
    ring = 4 if n > (1 << 21) else 32
    acts = sticks.ema_noise_device(ring, n, dev)
    h = C.c_void_p(); _lib.check(L.fpv_create(C.byref(cp), n, 0, C.byref(h)))
    rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
    pads = [0, 64, 256, 512, 1024, 2048 + 256, 4096 + 256, 8192 + 256, 16384 + 512, 65536 + 256, 333 * 64]
    big = torch.zeros(14 * (n + max(pads)) + 64, device=dev)
    res = {q: [] for q in pads}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(6):
        for q in pads:
            ld = n + q
            st = big[:14 * ld].view(14, ld)
            st.zero_(); st[2] = 10; st[3] = 1; st[6] = 1
            b = _lib.FpvBuffers(); b.state, b.ld, b.reward, b.done, b.action = st.data_ptr(), ld, rew.data_ptr(), done.data_ptr(), acts.data_ptr()
            reps = 16 if n > (1 << 21) else 8
            torch.cuda.synchronize(); e0.record()
            for _ in range(reps):
                _lib.check(L.fpv_rollout(h, C.byref(b), ring, n * 4, 0, None))
            e1.record(); torch.cuda.synchronize()
            if r: res[q].append(e0.elapsed_time(e1) * 1e3 / (reps * ring))
    for q in pads:
        med = statistics.median(res[q])
        print(f"n=2^{n.bit_length() - 1} ld = n + {q:6d} floats: median {med:8.2f} us  {133 * n / med / 1e3:7.1f} GB/s  (recommended ld = n + {int(L.fpv_recommended_ld(n)) - n})", flush=True)
    L.fpv_destroy(h)
    del big, acts
