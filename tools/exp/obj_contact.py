#!/usr/bin/env python3
"""Cost of the object_list pass of the single-step kernel when EVERY lane runs it (the wave-level cull cannot
help): 2^20 drones hovering 0.2 m above a Ground entry and 0.15 m above the top of a huge Target sphere, so the
per-motor distance pass of both objects runs in every wave (no motor is in contact: no force, no crash), against
the same world far away (everything culled) and the plain kernel.  A/B over builds in ONE process on shared
buffers; 32 steps after each reset, so the drones stay where they were put.

    python tools/exp/obj_contact.py [--lib prev=tools/exp/libfpv_f_prev.so]
"""
import argparse
import ctypes as C
import os
import statistics
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from fpyv_amd import _lib, load_params, sticks  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--lib", nargs="*", default=[])
a = ap.parse_args()
libs = {"shipped": os.path.join(REPO, "fpyv_amd", "libfpv_hip.so")}
for e in a.lib:
    k, v = e.split("=", 1)
    libs[k] = os.path.join(REPO, v)
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
p = load_params(fps=1000)
cp = _lib.pack_params(p)
n, k = a.n, 32
acts = sticks.ema_noise_device(k, n, dev)
acts[..., 3] = -0.646                                 # hover throttle: the drones stay put for the 32 steps
L, H = {}, {}
for name, path in libs.items():
    lib = C.CDLL(path)
    lib.fpv_create.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    lib.fpv_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]
    lib.fpv_recommended_ld.argtypes = [C.c_int64]
    lib.fpv_recommended_ld.restype = C.c_int64
    lib.fpv_last_error.restype = C.c_char_p
    h = C.c_void_p()
    assert lib.fpv_create(C.byref(cp), n, 0, C.byref(h)) == 0, lib.fpv_last_error()
    L[name], H[name] = lib, h
ld = int(L["shipped"].fpv_recommended_ld(n))
st = torch.zeros((14, ld), device=dev)
rew = torch.zeros(n, device=dev)
done = torch.zeros(n, dtype=torch.uint8, device=dev)
worlds = {
    "no objects": None,
    "4 objects, all culled": _lib.pack_objects([(2, 0, -6, 3, 0.8, 0), (1, 3, 0, 0, 1.0, 5.0), (1, -2, 2.5, 0, 0.6, 1.5), (0, 0, 0, -50.0, 0, 0)]),
    "ground + sphere pass in every lane": _lib.pack_objects([(2, 0, 0, -999.95, 1000.0, 0), (0, 0, 0, 0, 0, 0)]),
}
b = _lib.FpvBuffers()
b.state, b.ld, b.reward, b.done, b.action = st.data_ptr(), ld, rew.data_ptr(), done.data_ptr(), acts.data_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
fin = {}
for wname, world in worlds.items():
    b.objects = C.addressof(world) if world is not None else None
    z0 = 0.2 if "every lane" in wname else 10.0
    res = {name: [] for name in libs}
    for r in range(a.rounds):
        for name in libs:
            st.zero_()
            st[2] = z0
            st[6] = 1
            torch.cuda.synchronize()
            e0.record()
            rc = L[name].fpv_rollout(H[name], C.byref(b), k, n * 4, 0, None)
            assert rc == 0, L[name].fpv_last_error()
            e1.record()
            torch.cuda.synchronize()
            if r:
                res[name].append(e0.elapsed_time(e1) * 1e3 / k)
            fin[(wname, name)] = (st.clone(), int(done.sum()))
    for name in libs:
        same = bool(torch.equal(fin[(wname, name)][0], fin[(wname, "shipped")][0]))
        print(f"n={n} {wname:36s} {name:10s}: median {statistics.median(res[name]):7.2f} us/step  min {min(res[name]):7.2f}   "
              f"crashed {fin[(wname, name)][1]}  bitwise==shipped {same}", flush=True)
