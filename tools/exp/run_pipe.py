#!/usr/bin/env python3
import ctypes as C, os, statistics, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from fpyv_amd import _lib, load_params, sticks
so = os.path.join(HERE, "libfpv_exp.so")
torch.zeros(1, device="cuda:0")
L = C.CDLL(so)
L.exp_rollout_pipelined.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
p = load_params(fps=1000); cp = _lib.pack_params(p)
n = 1 << 20; ld = n + 256; ring = 32
acts = sticks.ema_noise_device(ring, n, dev)
st = torch.zeros((14, ld), device=dev); reward = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
def reset(): st.zero_(); st[2] = 10; st[3] = 1; st[6] = 1
res = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
finals = {}
for r in range(6):
    for S in (1, 2, 3, 4, 8):
        reset(); torch.cuda.synchronize(); e0.record()
        for rep in range(8):
            rc = L.exp_rollout_pipelined(C.byref(cp), st.data_ptr(), ld, acts.data_ptr(), n * 4, reward.data_ptr(), done.data_ptr(), n, ring, S, None)
            assert rc == 0
        e1.record(); torch.cuda.synchronize()
        if r: res.setdefault(S, []).append(e0.elapsed_time(e1) * 1e3 / (8 * ring))
        finals[S] = st[:, :n].clone()
for S, v in res.items():
    med = statistics.median(v)
    print(f"S={S}: {med:7.2f} us per full step  {133 * n / med / 1e3:7.1f} GB/s(alg)  equal_to_S1={bool(torch.equal(finals[S], finals[1]))}", flush=True)
