#!/usr/bin/env python3
"""A/B builds of libfpv_hip.so that differ by a -DFPV_EXP_* switch (csrc/fpv_exp.h) or a code-generation flag, in ONE process on
the SAME buffers, interleaved (cdna_hip_programming.md 5.4 rule 24).  Variants are built where they are timed - hipcc is on the
GPU box too - into gpurun_out/_variants/ (scratch: never pushed, never committed):

    python tools/ab_variants.py --build --only base w4 b256 && python tools/ab_variants.py --only base w4 b256 [--n 1048576]

The switches of rounds 1-5 whose questions are closed (non-temporal hints, tiled state, row skew, two-tier head order,
VGPR constants, double prefetch) left the sources in round 6; what each measured is in profiles/HISTORY.md.
"""
import argparse, ctypes as C, os, statistics, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OUT = os.path.join(REPO, "gpurun_out", "_variants")
sys.path.insert(0, REPO)
VARIANTS = {
    "base": [],
    "w4": ["-DFPV_EXP_STEP_WAVES=4"], "w5": ["-DFPV_EXP_STEP_WAVES=5"], "w7": ["-DFPV_EXP_STEP_WAVES=7"], "w8": ["-DFPV_EXP_STEP_WAVES=8"], "w44": ["-DFPV_EXP_STEP_WAVES=4,4"], "w55": ["-DFPV_EXP_STEP_WAVES=5,5"], "w33": ["-DFPV_EXP_STEP_WAVES=3,3"],   # occupancy of the step kernel
    "b64": ["-DFPV_EXP_BLOCK=64"], "b256": ["-DFPV_EXP_BLOCK=256"],       # drones per workgroup (the rotation's block, the XCD's share of a row: 256 B of 2 KiB / 1 KiB of 8 KiB)
    "pre12": ["PRELOAD=12"], "pre16": ["PRELOAD=16"], "pre8": ["PRELOAD=8"],       # kernel-argument dwords preloaded into SGPRs (shipped: 6 = state, ld, action; 12 reaches n_start)
}
ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--only", nargs="*", default=None)
ap.add_argument("--rotations", nargs="*", type=int, default=[-1], help="fpv_set_rotation values (drones; -1 automatic, 0 plain order) to time every variant at")
ap.add_argument("--states", type=int, default=1, help="time every variant on this many separately allocated state matrices (placement matters beyond the cache)")
a = ap.parse_args()
names = [k for k in VARIANTS if not a.only or k in a.only]
if a.build:
    for k in names:
        os.makedirs(OUT, exist_ok=True)
        out = os.path.join(OUT, f"libfpv_v_{k}.so")
        from __graft_entry__ import HIPCC_FLAGS
        flags, extra = list(HIPCC_FLAGS), [f for f in VARIANTS[k] if not f.startswith("PRELOAD=")]
        for f in VARIANTS[k]:
            if f.startswith("PRELOAD="):
                flags = [("-amdgpu-kernarg-preload-count=" + f.split("=")[1]) if x.startswith("-amdgpu-kernarg-preload-count=") else x for x in flags]
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, *extra, "-o", out, os.path.join(REPO, "fpyv_amd", "csrc", "fpv_hip.hip")], check=True)
        print("built", out)
    sys.exit(0)
import torch
from fpyv_amd import _lib, load_params, sticks
dev = torch.device("cuda:0"); torch.zeros(1, device=dev)
p = load_params(fps=1000, ceiling=100.0); cp = _lib.pack_params(p, auto_reset=True)
n = a.n; ring = 32 if n <= (1 << 21) else 4
acts = sticks.ema_noise_device(ring, n, dev)
L, H = {}, {}
for k in names:
    l = C.CDLL(os.path.join(OUT, f"libfpv_v_{k}.so"))
    l.fpv_create.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    l.fpv_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]
    l.fpv_recommended_ld.argtypes = [C.c_int64]; l.fpv_recommended_ld.restype = C.c_int64
    l.fpv_last_error.restype = C.c_char_p
    l.fpv_set_rotation.argtypes = [C.c_void_p, C.c_int64]
    h = C.c_void_p(); rc = l.fpv_create(C.byref(cp), n, 0, C.byref(h)); assert rc == 0, l.fpv_last_error()
    L[k], H[k] = l, h
ld = int(L[names[0]].fpv_recommended_ld(n))
rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
if a.states > 1:
    # the same variants on several state matrices of this process, freed and allocated again in between: one line per matrix
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    keep = []
    for j in range(a.states):
        if j % 2 == 1:
            torch.cuda.empty_cache()
        keep.append(torch.empty((5 + 13 * j) << 20, dtype=torch.uint8, device=dev))
        st = torch.zeros((14, ld), device=dev)
        b = _lib.FpvBuffers(); b.state, b.ld, b.reward, b.done, b.action = st.data_ptr(), ld, rew.data_ptr(), done.data_ptr(), acts.data_ptr()
        row = []
        for k in names:
            ts = []
            for r in range(3):
                st.zero_(); st[2] = 10; st[3] = 1; st[6] = 1
                assert L[k].fpv_rollout(H[k], C.byref(b), ring, n * 4, 0, None) == 0
                torch.cuda.synchronize(); e0.record()
                for rep in range(6):
                    assert L[k].fpv_rollout(H[k], C.byref(b), ring, n * 4, 0, None) == 0
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / (6 * ring))
            row.append(f"{k} {statistics.median(ts):7.2f}")
        print(f"state 0x{st.data_ptr():x}: " + "  ".join(row), flush=True)
        if j % 3 != 2:
            keep.append(st)
        del st
    sys.exit(0)
st = torch.zeros((14, ld), device=dev)
b = _lib.FpvBuffers(); b.state, b.ld, b.reward, b.done = st.data_ptr(), ld, rew.data_ptr(), done.data_ptr()
b.action = acts.data_ptr()
def reset(k="base"):
    st.zero_(); st[2] = 10; st[3] = 1; st[6] = 1
res = {(k, rot): [] for k in names for rot in a.rotations}; fin = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 8 if n <= (1 << 21) else 16
for r in range(a.rounds):
    for k in names:
        for rot in a.rotations:
            assert L[k].fpv_set_rotation(H[k], rot) == 0
            reset(k); torch.cuda.synchronize(); e0.record()
            for rep in range(reps):
                rc = L[k].fpv_rollout(H[k], C.byref(b), ring, n * 4, 0, None); assert rc == 0, L[k].fpv_last_error()
            e1.record(); torch.cuda.synchronize()
            if r: res[(k, rot)].append(e0.elapsed_time(e1) * 1e3 / (reps * ring))
        fin[k] = st.clone()
for k in names:
    for rot in a.rotations:
        med = statistics.median(res[(k, rot)])
        print(f"n={n} {k:12s} rotation {rot:8d}: median {med:8.3f} us  min {min(res[(k, rot)]):8.3f} us   {133 * n / med / 1e3:8.1f} GB/s   bitwise==base {bool(torch.equal(fin[k][:, :n], fin[names[0]][:, :n]))}", flush=True)
