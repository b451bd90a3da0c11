#!/usr/bin/env python3
"""Which row stride of the state matrix does a population want?  For each population: the launch time of the fp32 step kernel
(fpv_rollout, one C call per ring span) with the automatic rotation and in the plain order, for the eight 256-byte classes of the
stride (ld = a multiple of 512 floats + c * 64), for fpv_recommended_ld(n) and for the rule of rounds 1-4 (n rounded to 64,
1 KiB clear of a multiple of 8 KiB) - ONE allocation per population, every stride a view of it.

    python tools/row_stride_sweep.py 524288,1048576,1000000          # GPU box

profiles/r05_exp_row_stride_l2_sets.log holds the sweeps that fpv_recommended_ld's rule and the L2 set model (fpv_hip.hip
l2_set_overflow; tools/l2_set_model.py compares it with these logs) were made from."""
import sys, os, time, statistics, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpyv_amd import _lib, load_params, sticks
dev = torch.device("cuda:0"); torch.zeros(1, device=dev)
L = _lib.lib()
p = load_params(fps=1000, ceiling=100.0); cp = _lib.pack_params(p, auto_reset=True)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
period = int(os.environ.get("PERIOD", "512")); step = int(os.environ.get("STEP", "64"))
for n in [int(x) for x in sys.argv[1].split(",")]:
    ring = 32 if n <= (1 << 21) else 4
    acts = sticks.ema_noise_device(ring, n, dev)
    h = C.c_void_p(); assert L.fpv_create(C.byref(cp), n, 0, C.byref(h)) == 0
    rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
    base = (n + period - 1) // period * period
    rec = int(L.fpv_recommended_ld(n))
    old = (n + 63) // 64 * 64
    old += (256 - old % 2048) if old % 2048 < 256 else 0
    lds = [base + c * step for c in range(period // step)] + [rec, old]
    big = torch.zeros(14 * (max(lds) + period) + 64, device=dev)
    res = {}
    reps = max(2, (1 << 23) // n // ring * 2)
    for rnd in range(4):
        for ld in lds:
            st = big[:14 * ld].view(14, ld)
            b = _lib.FpvBuffers(); b.state, b.ld, b.reward, b.done, b.action = st.data_ptr(), ld, rew.data_ptr(), done.data_ptr(), acts.data_ptr()
            for rot in (-1, 0):
                L.fpv_set_rotation(h, rot)
                big.zero_(); st[2] = 10; st[3] = 1; st[6] = 1
                for rep in range(2): assert L.fpv_rollout(h, C.byref(b), ring, n * 4, 0, None) == 0
                torch.cuda.synchronize(); ev0.record()
                for rep in range(reps): assert L.fpv_rollout(h, C.byref(b), ring, n * 4, 0, None) == 0
                ev1.record(); torch.cuda.synchronize()
                if rnd: res.setdefault((ld, rot), []).append(ev0.elapsed_time(ev1) * 1e3 / (reps * ring))
    print(f"n={n:8d} (rec ld = n+{rec - n}, class {(rec % period) // step}; old n+{old - n})  auto: " + " ".join(f"{statistics.median(res[(ld, -1)]):6.2f}" for ld in lds[:-2]) +
          f" | rec {statistics.median(res[(rec, -1)]):6.2f} old {statistics.median(res[(old, -1)]):6.2f}" +
          "   plain: " + " ".join(f"{statistics.median(res[(ld, 0)]):6.2f}" for ld in lds[:-2]) + f" | rec {statistics.median(res[(rec, 0)]):6.2f} old {statistics.median(res[(old, 0)]):6.2f}", flush=True)
    L.fpv_destroy(h); del big, acts, rew, done; torch.cuda.empty_cache()
