#!/usr/bin/env python3
"""A/B builds of libfpv_hip.so (compiler flags and/or -DFPV_EXP_* macros), in ONE process on the SAME buffers,
interleaved (cdna_hip_programming.md 5.4 rule 24): times the single-step kernel (fpv_rollout = k launches) AND
the k-step kernel (fpv_step_n, one launch per ring span) of every build, and checks that every build leaves
bit-identical state.

    python tools/exp/ab_fused.py --build            # in the build container (hipcc cross-compiles)
    python tools/exp/ab_fused.py [--n 1048576]      # on the GPU box
"""
import argparse
import ctypes as C
import os
import statistics
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
BASE = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC"]   # variants add the SLP switch themselves
VARIANTS = {
    "shipped": None,                      # fpyv_amd/libfpv_hip.so as built by __graft_entry__.build()
    "slp": ["-fslp-vectorize"],
    "noslp": ["-fno-slp-vectorize"],
}
ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--only", nargs="*", default=None)
ap.add_argument("--extra", nargs="*", default=[], help="name=flag,flag,... additional variants")
ap.add_argument("--lib", nargs="*", default=[], help="name=path/to/lib.so: a ready-built library (e.g. an older commit) as a variant")
ap.add_argument("--geom", default="f32", choices=["f32", "kahan", "racerW", "racerD", "racerWC", "aos", "noise", "fp16"],
                help="which kernel family to time: the plain drone kernel, + Kahan rows, the Racer as written / omega*dt / components.PID, the AoS observation head")
a = ap.parse_args()
for e in a.extra:
    k, v = e.split("=", 1)
    VARIANTS[k] = [f for f in v.split(",") if f and f != "x"]        # "name=x": the library built earlier under that name
READY = {}
for e in a.lib:
    k, v = e.split("=", 1)
    READY[k] = os.path.join(REPO, v)
    VARIANTS[k] = None
names = [k for k in VARIANTS if not a.only or k in a.only]


def path_of(k):
    if k in READY:
        return READY[k]
    return os.path.join(REPO, "fpyv_amd", "libfpv_hip.so") if VARIANTS[k] is None else os.path.join(HERE, f"libfpv_f_{k}.so")


if a.build:
    for k in names:
        if VARIANTS[k] is None:
            continue
        subprocess.run(["/opt/rocm/bin/hipcc", *BASE, *VARIANTS[k], "-o", path_of(k),
                        os.path.join(REPO, "fpyv_amd", "csrc", "fpv_hip.hip")], check=True)
        print("built", path_of(k))
    sys.exit(0)

import torch  # noqa: E402
from fpyv_amd import _lib, load_params, sticks  # noqa: E402

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
p = load_params(fps=1000, ceiling=100.0)
if a.geom.startswith("racer"):
    import numpy as np
    p = p.replace(mode=1, racer_pid=np.asarray([[0.004, 0.02, 1e-6], [0.003, 0.01, 2e-6], [0.002, 0.005, 0.0]]),
                  racer_omega_dt=(a.geom == "racerD"), racer_pid_variant=int(a.geom == "racerWC"), ceiling=100.0)
cp = _lib.pack_params(p, auto_reset=True, stick_noise=(a.geom == "noise"), noise_seed=11, fp16_state=(a.geom == "fp16"))
n = a.n
ring = 32 if n <= (1 << 21) else 4
acts = sticks.ema_noise_device(ring, n, dev)
L, H = {}, {}
for k in names:
    lib = C.CDLL(path_of(k))
    lib.fpv_create.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    for f in (lib.fpv_rollout, lib.fpv_step_n):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]
    lib.fpv_reset.argtypes = [C.c_void_p, C.c_void_p] + [C.c_void_p] * 5
    lib.fpv_recommended_ld.argtypes = [C.c_int64]
    lib.fpv_recommended_ld.restype = C.c_int64
    lib.fpv_last_error.restype = C.c_char_p
    h = C.c_void_p()
    rc = lib.fpv_create(C.byref(cp), n, 0, C.byref(h))
    assert rc == 0, lib.fpv_last_error()
    L[k], H[k] = lib, h
ld = int(L[names[0]].fpv_recommended_ld(n))
st = torch.zeros((_lib.state_rows(int(p.mode)), ld), device=dev)
rew = torch.zeros(n, device=dev)
done = torch.zeros(n, dtype=torch.uint8, device=dev)
b = _lib.FpvBuffers()
b.state, b.ld, b.reward, b.done = st.data_ptr(), ld, rew.data_ptr(), done.data_ptr()
b.action = acts.data_ptr()
extra = None
if a.geom == "kahan":
    extra = torch.zeros((6, ld), device=dev)
    b.pos_comp = extra.data_ptr()
elif a.geom == "fp16":
    extra = torch.zeros(_lib.FPV_HALF_HALVES * ld, dtype=torch.float16, device=dev)
    b.state_h, b.rounding_seed = extra.data_ptr(), 5
elif a.geom == "noise":
    extra = torch.zeros((4, ld), device=dev)
    b.noise_state = extra.data_ptr()
    b.action = None                                  # pure in-kernel noise sticks
elif a.geom == "aos":
    extra = torch.zeros((n, _lib.FPV_OBS_AOS_DIM), device=dev)
    b.obs_aos = extra.data_ptr()


def reset():
    st.zero_()
    if a.geom == "fp16":
        extra.zero_()
    rc = L[names[0]].fpv_reset(H[names[0]], C.byref(b), None, None, None, None, None)
    assert rc == 0, L[names[0]].fpv_last_error()


e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for api in (("fpv_rollout",) if a.geom == "aos" else ("fpv_rollout", "fpv_step_n")):
    reps = (8 if n <= (1 << 21) else 16) * (6 if api == "fpv_step_n" else 1)      # ~6-8 ms per timing either way
    res = {k: [] for k in names}
    fin = {}
    for r in range(a.rounds):
        for k in names:
            fn = getattr(L[k], api)
            reset()
            torch.cuda.synchronize()
            e0.record()
            for rep in range(reps):
                rc = fn(H[k], C.byref(b), ring, 0 if a.geom == "noise" else n * 4, 0, None)
                assert rc == 0, L[k].fpv_last_error()
            e1.record()
            torch.cuda.synchronize()
            if r:
                res[k].append(e0.elapsed_time(e1) * 1e3 / (reps * ring))
            fin[k] = st.clone()
    for k in names:
        med = statistics.median(res[k])
        print(f"n={n} {a.geom:8s} {api:12s} {k:12s}: median {med:8.3f} us/step  min {min(res[k]):8.3f}   {n / med / 1e3:8.2f} G env-steps/s   "
              f"bitwise==first {bool(torch.equal(fin[k], fin[names[0]]))}", flush=True)
