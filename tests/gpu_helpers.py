"""Helpers shared by the GPU test files (tests/test_gpu_*.py): batch construction, golden replays."""
import numpy as np
import torch

from conftest import load_golden  # noqa: F401
from fpyv_amd import _lib, load_params, sticks  # noqa: F401
from oracle import lane_model, oracle  # noqa: F401
from parity import REL_TOL, assert_parity, soa_vs_oracle  # noqa: F401

DEV = "cuda:0"


def _drone_batch(p, n, **kw):
    from fpyv_amd.env import DroneBatch
    return DroneBatch(p, n, device=DEV, **kw)


def _run_golden(p, g, per_step_calls=False):
    acts = g["actions"]
    T, n = acts.shape[:2]
    env = _drone_batch(p, n)
    env.reset(position=g["init_position"], velocity=g["init_velocity"], ypr=g["init_ypr"])
    a = torch.from_numpy(acts).to(DEV)
    if per_step_calls:
        for t in range(T):
            env.step(a[t], wind_velocity_vector=g["wind"], object_list=[], return_imu=False)
    else:
        env.rollout(a, wind=g["wind"])
    torch.cuda.synchronize()
    return env


def _racer_replay(p, g):
    """Replay a Racer golden through RacerBatch, comparing at every snapshot; returns the worst errors."""
    from fpyv_amd.env import RacerBatch
    env = RacerBatch(p, 1, device=DEV)
    env.reset()
    acts = torch.from_numpy(g["actions"]).to(DEV)
    prev, worst = 0, dict(quat=0.0, pos=0.0, omega=0.0)
    for k, t in enumerate(np.asarray(g["snap_steps"]).reshape(-1)):
        env.rollout(acts[prev:int(t)].contiguous())
        prev = int(t)
        s = env.state.cpu().numpy()
        q = s[6:10, 0].astype(np.float64)
        x, y, z, w = g["quat_xyzw"][0, k]
        qr = np.array([w, x, y, z])
        q *= np.sign(q @ qr)
        pr = g["position"][0, k]
        worst["quat"] = max(worst["quat"], np.abs(q - qr).max())
        worst["pos"] = max(worst["pos"], np.abs(s[0:3, 0] - pr).max() / max(np.abs(pr).max(), 1e-3))
        worst["omega"] = max(worst["omega"], np.abs(s[10:13, 0].astype(np.float64) + s[20:23, 0] - g["omega"][0, k]).max())
    return env, worst


# ---- fpv_step_n: k steps in ONE launch, bit-identical to k single steps ---------------------------------
def _clone_batch_state(dst, src):
    for name in ("state", "state_h", "noise_state", "pos_comp", "ep_return", "ep_length", "last_return", "last_length"):
        a, b = getattr(dst, name, None), getattr(src, name, None)
        if a is not None:
            a.copy_(b)


def _two_host_threads_two_handles(params, devices, one_world):
    """Two host threads in ONE process, a handle and a communicator rank per thread, neither thread ever calling
    hipSetDevice itself.  one_world: both threads join ONE communicator of world size 2 (needs two GPUs); otherwise each
    thread has its own one-rank communicator (what a one-GPU box can run: the threading of the C ABI - thread-local error
    strings, RCCL opened under call_once - and the device guard are the same code)."""
    import ctypes as C
    import threading
    L = _lib.lib()
    idents = []
    for _ in range(1 if one_world else 2):
        ident = (C.c_uint8 * _lib.FPV_COMM_ID_BYTES)()
        _lib.check(L.fpv_comm_unique_id(ident))
        idents.append(ident)
    torch.cuda.set_device(0)
    n, k = 5000, 16
    words = (n + 63) // 64
    out, errors = {}, []
    p = params.replace(ceiling=10.0005)
    world = 2 if one_world else 1

    def rank_main(r):
        try:
            dev = torch.device("cuda", devices[r])
            seen = [torch.cuda.current_device()]                      # a fresh thread: device 0 is current, also for rank 1
            from fpyv_amd.env import DroneBatch
            env = DroneBatch(p, n, device=dev, with_done_bits=True, auto_reset=True, with_accel=False, drone_id_offset=r * n)
            env.reset()
            acts = torch.from_numpy(sticks.ema_noise(k, range(r * n, (r + 1) * n), seed=4)).to(dev)
            acts[..., 3] = 1.0                                          # full throttle: through the ceiling within a few steps
            if r == 1:
                acts[:, ::3, 3] = -0.9                                  # every third drone of rank 1 sinks instead: the ranks' masks differ
            comm = C.c_void_p()
            _lib.check(L.fpv_comm_create(idents[0 if one_world else r], world, r if one_world else 0, devices[r], C.byref(comm)))   # collective
            seen.append(torch.cuda.current_device())
            ws, rk, ver = C.c_int(-1), C.c_int(-1), C.c_int(-1)
            _lib.check(L.fpv_comm_info(comm, C.byref(ws), C.byref(rk), C.byref(ver)))
            bucket = torch.zeros((k, words), dtype=torch.int64, device=dev)
            dones = torch.zeros((k, n), dtype=torch.uint8, device=dev)
            env.set_done_bits_target(bucket, stride_words=words)
            env.rollout(acts, dones=dones)                              # ONE launch on this rank's GPU writes all k mask rows
            seen.append(torch.cuda.current_device())
            gathered = torch.full((world, k, words), -1, dtype=torch.int64, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(L.fpv_allgather_done(comm, bucket.data_ptr(), gathered.data_ptr(), k * words, stream))
            seen.append(torch.cuda.current_device())
            torch.cuda.synchronize(dev)
            assert L.fpv_allgather_done(comm, None, gathered.data_ptr(), words, stream) == -1
            assert b"null argument" in L.fpv_last_error()            # this thread's own message (thread-local)
            out[r] = dict(bucket=bucket.cpu(), gathered=gathered.cpu(), dones=dones.cpu(), seen=seen, info=(ws.value, rk.value, ver.value),
                          state_device=env.state.device.index)
            L.fpv_comm_destroy(comm)
            seen.append(torch.cuda.current_device())
        except Exception as e:      # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in RCCL"
    assert not errors, errors
    assert torch.cuda.current_device() == 0
    whole = torch.stack([out[0]["bucket"], out[1]["bucket"]])
    for r in range(2):
        o = out[r]
        assert o["info"][:2] == ((2, r) if one_world else (1, 0)) and o["info"][2] >= 20000
        assert o["seen"] == [0, 0, 0, 0, 0], f"rank {r}: an fpv_* call left the caller's current device changed: {o['seen']}"
        assert o["state_device"] == devices[r]
        assert torch.equal(o["gathered"], whole if one_world else whole[r:r + 1]), f"rank {r}: gathered masks != concatenation of the ranks' buckets"
        bits = ((o["bucket"].numpy().view(np.uint64)[:, :, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).reshape(k, -1)[:, :n]
        assert np.array_equal(bits.astype(np.uint8), o["dones"].numpy()) and bits.any()
    assert not torch.equal(out[0]["bucket"], out[1]["bucket"])
