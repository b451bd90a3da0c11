#!/usr/bin/env python3
"""A/B two builds of libfpv_hip.so (same ABI) in ONE process on the SAME buffers, interleaved."""
import ctypes as C, os, statistics, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from fpyv_amd import _lib, load_params, sticks
libs = {"new": _lib.LIB_PATH, "old": os.path.join(HERE, "libfpv_old.so"), "addr64": os.path.join(HERE, "libfpv_addr64.so")}
dev = torch.device("cuda:0"); torch.zeros(1, device=dev)
p = load_params(fps=1000); cp = _lib.pack_params(p)
n = 1 << 20; ring = 32
acts = sticks.ema_noise_device(ring, n, dev)
L, H = {}, {}
for k, path in libs.items():
    l = C.CDLL(path)
    l.fpv_create.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    l.fpv_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]
    l.fpv_recommended_ld.argtypes = [C.c_int64]; l.fpv_recommended_ld.restype = C.c_int64
    h = C.c_void_p(); assert l.fpv_create(C.byref(cp), n, 0, C.byref(h)) == 0
    L[k], H[k] = l, h
ld = int(L["new"].fpv_recommended_ld(n))
st = torch.zeros((14, ld), device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
b = _lib.FpvBuffers(); b.state, b.ld, b.reward, b.done = st.data_ptr(), ld, rew.data_ptr(), done.data_ptr()
b.action = acts.data_ptr()
def reset(): st.zero_(); st[2] = 10; st[3] = 1; st[6] = 1
res = {k: [] for k in libs}; fin = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(8):
    for k in libs:
        reset(); torch.cuda.synchronize(); e0.record()
        for rep in range(8):
            assert L[k].fpv_rollout(H[k], C.byref(b), ring, n * 4, 0, None) == 0
        e1.record(); torch.cuda.synchronize()
        if r: res[k].append(e0.elapsed_time(e1) * 1e3 / (8 * ring))
        fin[k] = st.clone()
for k in libs:
    print(f"{k}: median {statistics.median(res[k]):.3f} us  min {min(res[k]):.3f} us")
print("bitwise equal:", bool(torch.equal(fin["new"], fin["old"])), bool(torch.equal(fin["new"], fin["addr64"])))
