"""Host cost of one step at a launch-bound population (4096 drones): where the microseconds between two launches go."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from fpyv_amd import load_params  # noqa: E402
from fpyv_amd.env import DroneBatch, FpvVecEnv  # noqa: E402

dev = torch.device("cuda", 0)
n, K = 4096, 20000
e = DroneBatch(load_params(fps=1000, ceiling=100.0), n, device=dev, auto_reset=True, with_accel=False)
e.reset()
a = torch.zeros((n, 4), device=dev)
ring = torch.zeros((64, n, 4), device=dev)
ve = FpvVecEnv(load_params(fps=1000, ceiling=100.0), num_envs=n, device=dev)
ve.reset()
from fpyv_amd.pid import PID  # noqa: E402
pid = PID(0.1, 2.0, 0.05, dt=1e-3, integral_clip=100.0, min_output=1.5, max_output=80.0, num_envs=n, device=dev)
cur, tgt = torch.rand(n, device=dev), torch.rand(n, device=dev)
Rg = torch.eye(3, device=dev).expand(n, 3, 3).contiguous()
fg = torch.full((n,), 7.0, device=dev)
L, h, ref, stream = e._L, e._handle, e._buf_ref, e._stream()
e._step_raw(a)


def t(fn, label):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        fn(i)
    torch.cuda.synchronize()
    print(f"{label:62s}: {(time.perf_counter() - t0) * 1e6 / K:6.2f} us per step", flush=True)


for _ in range(2):
    t(lambda i: L.fpv_step(h, ref, stream), "ctypes fpv_step(handle, buffers, stream), nothing else")
    t(lambda i: e._step_raw(a), "DroneBatch._step_raw(same tensor)")
    t(lambda i: e.step(a, return_imu=False), "DroneBatch.step(same tensor, return_imu=False)")
    t(lambda i: e.step(ring[i & 63], return_imu=False), "DroneBatch.step(ring[i], ...): a fresh view object per step")
    t(lambda i: e.step(a), "DroneBatch.step(same tensor): with the reference's return triple")
    t(lambda i: ve.step(a), "FpvVecEnv.step(same tensor) -> (obs, reward, done, info)")
    t(lambda i: pid(cur, 1.5), "PID.__call__(current tensor, scalar target)")
    t(lambda i: pid(cur, tgt), "PID.__call__(current tensor, target tensor)")
    t(lambda i: e.step(a, rotation_matrix=Rg, thrust_force=fg, return_imu=False), "DroneBatch.step(..., rotation_matrix=[n,3,3], thrust_force=[n])")
    t(lambda i: e.rotation_matrix, "DroneBatch.rotation_matrix (no step)")
    t(lambda i: e.euler_angles, "DroneBatch.euler_angles (no step)")
    print()
