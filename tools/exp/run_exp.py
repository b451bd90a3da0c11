#!/usr/bin/env python3
"""Interleaved timing of tools/exp/exp_kernels.hip variants (one process, rounds interleaved).
All variants run on the SAME buffers (timings depend on buffer placement, so separate allocations
per variant confound the comparison); --pads sweeps the SoA row stride ld = n + pad floats."""
import argparse, ctypes as C, json, os, statistics, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from fpyv_amd import _lib, load_params, sticks

def build():
    so = os.path.join(HERE, "libfpv_exp.so")
    src = os.path.join(HERE, "exp_kernels.hip")
    if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-shared", "-fPIC",
                        "-o", so, src], check=True)
    return so

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--launches", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--ring", type=int, default=32)
    ap.add_argument("--variants", type=str, default="0,100")
    ap.add_argument("--pads", type=str, default="0")
    ap.add_argument("--offset", type=int, default=0, help="floats to shift the state base by (multiple of 4)")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    so = build()
    torch.zeros(1, device="cuda:0")
    L = C.CDLL(so)
    L.exp_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
    dev = torch.device("cuda:0")
    p = load_params(fps=1000)
    cp = _lib.pack_params(p)
    n = a.n
    acts = sticks.ema_noise_device(a.ring, n, dev)
    variants = []
    for tok in a.variants.split(","):
        v, _, g = tok.partition(":")
        variants.append((int(v), int(g) if g else 0))
    pads = [int(x) for x in a.pads.split(",")]
    maxld = n + max(pads)
    backing = torch.zeros(max(14 * maxld, 16 * n) + a.offset + 64, device=dev)
    reward = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
    def view(pad):
        ld = n + pad
        return backing[a.offset:a.offset + 14 * ld].view(14, ld), ld
    def reset(st):
        st.zero_(); st[2] = 10; st[3] = 1; st[6] = 1
    def reset_aos():
        v = backing[a.offset:a.offset + 16 * n].view(n, 16)
        v.zero_(); v[:, 2] = 10; v[:, 3] = 1; v[:, 6] = 1
    def reset_grp(pad):                       # variants 500+: four group rows (see k_grp): pz = 10, vx = 1, qw = 1
        ld = n + pad
        flat = backing[a.offset:a.offset + 14 * ld]
        flat.zero_()
        g0 = flat[:4 * ld].view(ld, 4); g1 = flat[4 * ld:8 * ld].view(ld, 4)
        g0[:, 2] = 10; g0[:, 3] = 1; g1[:, 2] = 1
    cases = [(v, pad) for pad in pads for v in variants]
    times = {c: [] for c in cases}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    finals = {}
    for r in range(a.rounds + 1):
        for c in cases:
            v, pad = c
            st, ld = view(pad)
            reset(st)
            if 300 <= v[0] < 400: reset_aos()
            if v[0] >= 500: reset_grp(pad)
            torch.cuda.synchronize(); e0.record()
            for t in range(a.launches):
                rc = L.exp_step(C.byref(cp), st.data_ptr(), ld, acts[t % a.ring].data_ptr(), reward.data_ptr(), done.data_ptr(), n, v[0], v[1], None)
                assert rc == 0, (v, rc)
            e1.record(); torch.cuda.synchronize()
            if r: times[c].append(e0.elapsed_time(e1) * 1e3 / a.launches)
            if a.check and r == a.rounds:
                if v[0] >= 500:
                    ld_ = n + pad
                    flat = backing[a.offset:a.offset + 14 * ld_]
                    finals[c] = torch.cat([flat[:4 * ld_].view(ld_, 4)[:n], flat[4 * ld_:8 * ld_].view(ld_, 4)[:n], flat[8 * ld_:12 * ld_].view(ld_, 4)[:n],
                                           flat[12 * ld_:14 * ld_].view(ld_, 2)[:n]], dim=1).t().clone()
                else:
                    finals[c] = (backing[a.offset:a.offset + 16 * n].view(n, 16)[:, :14].t().clone() if v[0] >= 300 else st[:, :n].clone())
    res = []
    for c in cases:
        v, pad = c
        med, mn = statistics.median(times[c]), min(times[c])
        print(f"variant {v[0]:3d} grid {v[1]:5d} pad {pad:6d}: median {med:7.2f} us min {mn:7.2f} us  {133 * n / med / 1e3:7.1f} GB/s", flush=True)
        res.append({"variant": v[0], "grid": v[1], "pad": pad, "median_us": med, "min_us": mn})
    if a.check:
        ref = finals[cases[0]]
        bad = [c for c in cases[1:] if not torch.equal(finals[c], ref)]
        print("state mismatches vs first case:", bad)
    if a.out: json.dump(res, open(a.out, "w"), indent=1)

if __name__ == "__main__":
    main()
