#!/usr/bin/env python3
"""A/B of the register-resident env server probe (tools/env_server_probe.hip) against the shipped closed loop.

    python tools/env_server_ab.py --build                       # build container: tools/_variants/libfpv_server_probe.so
    python tools/env_server_ab.py [--n 1048576] [--dpl 4] [--steps 300] [--gate kernel|streamop] [--hidden 0]   # GPU box

closed loop as shipped:   for t: sticks = policy(obs);  env.step(sticks)            (one stream, K x (policy kernels + 1 launch))
env server:               ONE persistent kernel holds the drones in registers; per step the policy's stream is gated on `ready >= t`
                          (a one-wave gate kernel, or hipStreamWaitValue32), runs the SAME policy kernels, and rings `bell = t + 1`.
Both run the same policy on the same initial state: the final states must be bit-identical.  Every wait in the probe has an
iteration cap and a wall-clock cap (--wait-cap-ms); the run is wrapped in `timeout` by the caller."""
import argparse
import ctypes as C
import os
import statistics
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
LIB = os.path.join(HERE, "_variants", "libfpv_server_probe.so")
LIB_BYPASS = os.path.join(HERE, "_variants", "libfpv_server_probe_bypass.so")

ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--dpl", type=int, default=4, help="drones per lane of the server kernel")
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--gate", choices=["kernel", "streamop"], default="kernel")
ap.add_argument("--hidden", type=int, default=0, help="0: linear 13 -> 4 policy; > 0: one hidden layer of that width")
ap.add_argument("--wait-cap-ms", type=float, default=200.0)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--poll-sleep", type=int, default=1, help="0 / 1 / 2: s_sleep 1 / 8 / 32 between polls of the doorbell")
ap.add_argument("--policy", choices=["torch", "none", "kernel"], default="torch",
                help="none: the policy's stream only rings the bell (what the env side alone costs); kernel: the linear policy as ONE hand-written kernel - "
                     "launched per step in the shipped loop, PERSISTENT beside the server (no launch at all between steps: what a fused policy would be)")
ap.add_argument("--policy-waves", type=int, default=1024)
ap.add_argument("--bypass", action="store_true", help="the build whose exchanged bytes use write-through stores / sc1 loads instead of cache write-back + invalidate fences")
a = ap.parse_args()
if a.build:
    from __graft_entry__ import HIPCC_FLAGS
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if "kernarg-preload" not in f and f != "-mllvm"]
    subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-Rpass-analysis=kernel-resource-usage", "-o", LIB, os.path.join(HERE, "env_server_probe.hip")], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-DSRV_BYPASS=1", "-o", LIB_BYPASS, os.path.join(HERE, "env_server_probe.hip")], check=True)
    print("built", LIB, LIB_BYPASS)
    sys.exit(0)

import torch  # noqa: E402

from fpyv_amd import _lib, load_params  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

dev = torch.device("cuda", 0)
n, K = a.n, a.steps
params = load_params(fps=1000, ceiling=100.0)
S = C.CDLL(LIB_BYPASS if a.bypass else LIB)
S.srv_last_error.restype = C.c_char_p
vp = C.c_void_p
S.srv_launch.argtypes = [vp, C.c_int64, C.c_int, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_double, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
S.srv_gate.argtypes = [vp, C.c_uint32, vp, C.c_double, vp]
S.srv_set_word.argtypes = [vp, C.c_uint32, vp]
S.srv_stream_wait_ge.argtypes = [vp, vp, C.c_uint32]
S.srv_stream_write.argtypes = [vp, vp, C.c_uint32]
S.srv_signal_alloc.argtypes = [C.POINTER(vp)]
S.srv_policy_once.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_int64, vp]
S.srv_policy_persistent.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp, vp, vp, C.c_double, vp]

torch.manual_seed(3)
if a.hidden:
    W1 = torch.randn(13, a.hidden, device=dev) * 0.05
    W2 = torch.randn(a.hidden, 4, device=dev) * 0.2
bias = torch.tensor([0.0, 0.0, 0.0, 0.4], device=dev)
Wt = torch.randn(13, 4, device=dev) * 0.02


Wh = (C.c_float * 52)(*[float(x) for x in Wt.cpu().reshape(-1)])
bh = (C.c_float * 4)(*[float(x) for x in bias.cpu()])


def policy(obs, out, state_ptr=None):
    """sticks = tanh(MLP(obs)) + bias into `out` [n, 4]; the same kernels in both arms"""
    if a.policy == "none":
        return
    if a.policy == "kernel":
        assert S.srv_policy_once(state_ptr, ld, out.data_ptr(), Wh, bh, n, torch.cuda.current_stream().cuda_stream) == 0
        return
    h = torch.tanh(obs @ W1) @ W2 if a.hidden else obs @ Wt
    torch.add(torch.tanh(h), bias, out=out)


env = DroneBatch(params, n, device=dev, auto_reset=True, with_accel=False)
env.reset()
torch.cuda.synchronize()
state0 = env.state.clone()
ld = env.ld
acts = torch.zeros((n, 4), device=dev)


def closed_loop():
    env.state.copy_(state0)
    obs = env.state[:13, :n].t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for t in range(K):
        policy(obs, acts, env.state.data_ptr())
        env.step(acts, return_imu=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / K, (time.perf_counter() - t0) * 1e6 / K


s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
srv_state = torch.zeros_like(state0)
reward = torch.zeros(n, device=dev)
done = torch.zeros(n, dtype=torch.uint8, device=dev)
words = torch.zeros(16, dtype=torch.int32, device=dev)          # [0] bell [4] ready [8] arrive [12] abort (separate 16-byte slots)
if a.gate == "streamop":
    bell_p, ready_p = vp(), vp()
    assert S.srv_signal_alloc(C.byref(bell_p)) == 0 and S.srv_signal_alloc(C.byref(ready_p)) == 0
    bell, ready = bell_p.value, ready_p.value
else:
    bell, ready = words.data_ptr(), words.data_ptr() + 16
arrive, abort_w = words.data_ptr() + 32, words.data_ptr() + 48
cp = _lib.pack_params(params, auto_reset=True)


def server_loop():
    srv_state.copy_(state0)
    words.zero_()
    if a.gate == "streamop":
        for p in (bell, ready):
            assert S.srv_stream_write(None, p, 0) == 0
    obs = srv_state[:13, :n].t()
    torch.cuda.synchronize()
    wpc, lim = C.c_int(), C.c_int()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(s2):
        e0.record()
    rc = S.srv_launch(C.byref(cp), n, a.dpl, srv_state.data_ptr(), ld, acts.data_ptr(), reward.data_ptr(), done.data_ptr(), bell, ready, arrive, abort_w,
                      K, a.wait_cap_ms, None, s1.cuda_stream, C.byref(wpc), C.byref(lim), a.poll_sleep)
    if rc != 0:
        raise SystemExit("srv_launch: " + S.srv_last_error().decode())
    if a.policy == "kernel":
        # the persistent pair: no launch between steps at all
        assert S.srv_policy_persistent(srv_state.data_ptr(), ld, acts.data_ptr(), Wh, bh, n, a.policy_waves, K, bell, ready, words.data_ptr() + 40, abort_w,
                                       a.wait_cap_ms, s2.cuda_stream) == 0, S.srv_last_error()
    with torch.cuda.stream(s2):
        for t in range(K if a.policy != "kernel" else 0):
            if t:
                if a.gate == "streamop":
                    assert S.srv_stream_wait_ge(s2.cuda_stream, ready, t) == 0
                else:
                    assert S.srv_gate(ready, t, abort_w, a.wait_cap_ms, s2.cuda_stream) == 0
            policy(obs, acts, srv_state.data_ptr())
            if a.gate == "streamop":
                assert S.srv_stream_write(s2.cuda_stream, bell, t + 1) == 0
            else:
                assert S.srv_set_word(bell, t + 1, s2.cuda_stream) == 0
    host_us = (time.perf_counter() - t0) * 1e6 / K
    s1.synchronize()
    with torch.cuda.stream(s2):
        e1.record()
    torch.cuda.synchronize()
    ab = int(words[12].item())
    return e0.elapsed_time(e1) * 1e3 / K, (time.perf_counter() - t0) * 1e6 / K, host_us, ab, wpc.value, lim.value


print(f"n = {n}, {K} steps, drones per lane {a.dpl}, {'bypass' if a.bypass else 'fences'}, policy waves {a.policy_waves}, gate {a.gate}, poll sleep {a.poll_sleep}, policy {a.policy} hidden {a.hidden}; can_stream_wait = {S.srv_can_stream_wait()}", flush=True)
for _ in range(2):
    closed_loop()
base = [closed_loop() for _ in range(a.rounds)]
final_base = env.state.clone()
print("closed loop as shipped : " + "  ".join(f"{e:.2f} us/step (wall {w:.2f})" for e, w in base), flush=True)
srv = []
for r in range(a.rounds + 1):
    e, w, h, ab, wpc, lim = server_loop()
    if ab:
        print(f"env server: ABORTED (abort word {ab}: a wait hit its cap) after round {r}; waves per CU {wpc}, resident limit {lim}", flush=True)
        break
    if r:
        srv.append((e, w, h))
if srv:
    print(f"env server ({a.gate:8s})  : " + "  ".join(f"{e:.2f} us/step (wall {w:.2f}, host enqueue {h:.2f})" for e, w, h in srv) + f"   [{wpc} waves per CU, resident limit {lim} waves]", flush=True)
    same = torch.equal(final_base[:, :n], srv_state[:, :n])
    print(f"final states bit-identical: {same}", flush=True)
    be, se = statistics.median(x[0] for x in base), statistics.median(x[0] for x in srv)
    print(f"median: closed loop {be:.2f} us/step = {n / be / 1e3:.2f} G env-steps/s;  env server {se:.2f} us/step = {n / se / 1e3:.2f} G env-steps/s;  ratio {be / se:.2f}x", flush=True)
