"""ctypes front-end of oracle/lane_model.cpp - the HIP kernel's per-lane fp32 arithmetic on the host.

TEST INFRASTRUCTURE ONLY (see lane_model.cpp): used by tests to bound fp32-vs-float64 error on the
CPU and to assert bit-exactness of the gfx950 kernel on the GPU box.  Never imported by fpyv_amd.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

from fpyv_amd import _lib as abi
from . import oracle as _oracle

_L = None


def lib() -> C.CDLL:
    global _L
    if _L is None:
        _oracle.build()
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libfpv_lane_model.so")
        if not os.path.isfile(path):
            _oracle.build(force=True)
        _L = C.CDLL(path)
        fp, u8 = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        _L.fpvl_run.argtypes = [C.POINTER(abi.FpvParams), C.c_int64, C.c_int, fp, C.c_int64, fp, C.c_int, fp, fp, u8, fp]
        _L.fpvl_run.restype = C.c_int
    return _L


def return_matrices(state: np.ndarray, n: int):
    """(rt [n,3,3], gyro [n,3,3]) of Drone.step's return triple from a [14, ld] fp32 state, by the kernel's own
    instructions (fpv_return_matrices, csrc/fpv_math.h)."""
    L = lib()
    L.fpvl_return_matrices.argtypes = [C.c_void_p] * 4
    L.fpvl_return_matrices.restype = None
    rt, gy = np.zeros((n, 9), dtype=np.float32), np.zeros((n, 9), dtype=np.float32)
    for i in range(n):
        q = np.ascontiguousarray(state[6:10, i], dtype=np.float32)
        r = np.ascontiguousarray(state[10:13, i], dtype=np.float32)
        L.fpvl_return_matrices(q.ctypes.data, r.ctypes.data, rt[i].ctypes.data, gy[i].ctypes.data)
    return rt.reshape(n, 3, 3), gy.reshape(n, 3, 3)


def quat_from_rpy_deg(roll: float, pitch: float, yaw: float) -> np.ndarray:
    """The reset kernel's own fp32 attitude for a per-drone ypr argument (fpv_quat_from_rpy_deg, csrc/fpv_math.h)."""
    L = lib()
    L.fpvl_quat_from_rpy_deg.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.fpvl_quat_from_rpy_deg.restype = None
    q = np.zeros(4, dtype=np.float32)
    L.fpvl_quat_from_rpy_deg(float(np.float32(roll)), float(np.float32(pitch)), float(np.float32(yaw)), q.ctypes.data)
    return q


def initial_state(p, n: int, position=None, velocity=None, ypr_deg=None, ld: Optional[int] = None,
                  as_reset_kernel: bool = False) -> np.ndarray:
    """SoA [rows, ld] fp32 state after a reset, built the way the float64 side builds it (host
    double math, rounded once) - used as the common starting point of parity runs.  `as_reset_kernel=True`
    forms the attitude of a per-drone `ypr_deg` with the reset kernel's own fp32 instructions instead
    (fpv_quat_from_rpy_deg): what fpv_reset leaves in the state, bit for bit."""
    from fpyv_amd.params import ypr_to_quat
    rows = abi.state_rows(int(p.mode))
    ld = ld or (n + 63) // 64 * 64
    s = np.zeros((rows, ld), dtype=np.float32)
    if int(p.mode) == abi.FPV_MODE_DRONE:
        pos = np.broadcast_to(np.asarray(p.init_position if position is None else position, float), (n, 3))
        vel = np.broadcast_to(np.asarray(p.init_velocity if velocity is None else velocity, float), (n, 3))
        ang = np.broadcast_to(np.asarray(p.init_orientation_deg if ypr_deg is None else ypr_deg, float), (n, 3))
        s[0:3, :n] = pos.T
        s[3:6, :n] = vel.T
        if as_reset_kernel and ypr_deg is not None:
            s[6:10, :n] = np.stack([quat_from_rpy_deg(*a) for a in ang]).T
        else:
            s[6:10, :n] = np.stack([ypr_to_quat(*a) for a in ang]).T
    else:
        s[abi.QW, :n] = 1.0
        s[abi.R_FIRST, :n] = 1.0
    return s


def set_pos_comp(comp: Optional[np.ndarray]) -> None:
    """[6, ld] fp32 Kahan compensation rows (p, v) for subsequent run() calls (same ld as the state); None = plain sums."""
    L = lib()
    L.fpvl_set_pos_comp.argtypes = [C.c_void_p]
    L.fpvl_set_pos_comp(None if comp is None else comp.ctypes.data)


_override_keep = None


def set_override(rotations=None, thrust_forces=None) -> None:
    """Guidance override for subsequent run() calls: rotations [n,3,3] fp32 and thrust forces [n] (NaN = not
    overridden), applied on every step of the call; None clears it."""
    global _override_keep
    L = lib()
    L.fpvl_set_override.argtypes = [C.c_void_p, C.c_void_p]
    if rotations is None:
        _override_keep = None
        L.fpvl_set_override(None, None)
        return
    r = np.ascontiguousarray(rotations, dtype=np.float32).reshape(-1, 9)
    f = np.ascontiguousarray(thrust_forces, dtype=np.float32).reshape(-1)
    assert r.shape[0] == f.shape[0]
    _override_keep = (r, f)
    L.fpvl_set_override(r.ctypes.data, f.ctypes.data)


def set_general_motors(on: bool) -> None:
    """True: subsequent run() calls evaluate the ground flag from all four motor heights even for the square X
    frame (the shortcut's unit test compares both)."""
    lib().fpvl_set_general_motors(int(bool(on)))


def set_objects(rows) -> None:
    """object_list (rows of (type, x, y, z, radius, height)) for subsequent run() calls; () clears it."""
    L = lib()
    L.fpvl_set_objects.argtypes = [C.c_void_p]
    if rows:
        t = abi.pack_objects(rows)
        L.fpvl_set_objects(C.addressof(t))
    else:
        L.fpvl_set_objects(None)


def run(p, state: np.ndarray, actions: np.ndarray, steps: Optional[int] = None, wind=(0.0, 0.0, 0.0),
        n: Optional[int] = None, auto_reset: bool = False) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Advance the SoA fp32 `state` in place.  Returns (state, accel [3,ld], done [n], reward [n])."""
    assert state.dtype == np.float32 and state.flags.c_contiguous
    ld = state.shape[1]
    actions = np.ascontiguousarray(actions, dtype=np.float32)
    per_step = actions.ndim == 3
    n = n if n is not None else actions.shape[-2]
    if per_step:
        steps = actions.shape[0] if steps is None else steps
    accel = np.zeros((3, ld), dtype=np.float32)
    done = np.zeros(n, dtype=np.uint8)
    reward = np.zeros(n, dtype=np.float32)
    w = np.asarray(wind, dtype=np.float32)
    cp = abi.pack_params(p, auto_reset=auto_reset)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    rc = lib().fpvl_run(C.byref(cp), n, steps, fp(state), ld, fp(actions), int(per_step), fp(w), fp(accel),
                        done.ctypes.data_as(C.POINTER(C.c_uint8)), fp(reward))
    if rc != 0:
        raise RuntimeError(f"fpvl_run failed with {rc}")
    return state, accel, done, reward


def run_h(p, pos: np.ndarray, sh: np.ndarray, actions: np.ndarray, steps: Optional[int] = None,
          wind=(0.0, 0.0, 0.0), seed0: int = 0, n: Optional[int] = None, auto_reset: bool = False, step0: int = 0,
          drone_id_offset: int = 0):
    """fp16-storage variant: pos [3, ld] float32 and sh [11 * ld] uint16 (binary16 bits in the layout of
    fpv_abi.h: five half2 pair rows, then one row of thrust halves), advanced in place exactly like
    fpv_drone_step_h_kernel; step t rounds with fpv_round_seed(seed0, step0 + t) (seed0 = the buffer's rounding_seed,
    step0 = the handle's 64-bit step counter; for step0 + t < 2^32 that is seed0 + step0 + t).  Returns (done [n], reward [n])."""
    L = lib()
    if not hasattr(L, "_h_ready"):
        L.fpvl_run_h.argtypes = [C.POINTER(abi.FpvParams), C.c_int64, C.c_int, C.POINTER(C.c_float),
                                 C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_int,
                                 C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.c_uint64]
        L.fpvl_run_h.restype = C.c_int
        L.fpvl_f32_to_f16.argtypes = [C.c_float, C.c_uint32, C.c_int]
        L.fpvl_f32_to_f16.restype = C.c_uint16
        L.fpvl_f16_to_f32.argtypes = [C.c_uint16]
        L.fpvl_f16_to_f32.restype = C.c_float
        L._h_ready = True
    assert pos.dtype == np.float32 and sh.dtype == np.uint16 and pos.flags.c_contiguous and sh.flags.c_contiguous
    ld = pos.shape[1]
    assert sh.shape == (abi.FPV_HALF_HALVES * ld,)
    actions = np.ascontiguousarray(actions, dtype=np.float32)
    per_step = actions.ndim == 3
    n = n if n is not None else actions.shape[-2]
    if per_step:
        steps = actions.shape[0] if steps is None else steps
    done = np.zeros(n, dtype=np.uint8)
    reward = np.zeros(n, dtype=np.float32)
    w = np.asarray(wind, dtype=np.float32)
    cp = abi.pack_params(p, auto_reset=auto_reset, fp16_state=True, drone_id_offset=drone_id_offset)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    rc = L.fpvl_run_h(C.byref(cp), n, steps, fp(pos), sh.ctypes.data, ld, fp(actions),
                      int(per_step), fp(w), int(seed0) & 0xFFFFFFFF, done.ctypes.data_as(C.POINTER(C.c_uint8)), fp(reward),
                      int(step0) & (2 ** 64 - 1))
    if rc != 0:
        raise RuntimeError(f"fpvl_run_h failed with {rc}")
    return done, reward


def f32_to_f16_rtz(x: np.ndarray) -> np.ndarray:
    """binary16 bit patterns of fp32 values rounded TOWARD ZERO (host side of v_cvt_pkrtz_f16_f32)."""
    L = lib()
    L.fpvl_f32_to_f16_rtz.argtypes = [C.c_float]
    L.fpvl_f32_to_f16_rtz.restype = C.c_uint16
    x = np.asarray(x, dtype=np.float32)
    return np.array([L.fpvl_f32_to_f16_rtz(float(v)) for v in x.reshape(-1)], dtype=np.uint16).reshape(x.shape)


def pack_state(state14, seed: int, drone: int) -> np.ndarray:
    """[6] uint32: the five half2 pair words and the thrust half the fp16 kernels store for one drone state
    (14 fp32 values in row order), rounding seed `seed` (= fpv_round_seed(rounding_seed, step)) and lane `drone`."""
    L = lib()
    L.fpvl_pack_state.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    st = np.ascontiguousarray(state14, dtype=np.float32)
    out = np.zeros(6, dtype=np.uint32)
    L.fpvl_pack_state(st.ctypes.data_as(C.POINTER(C.c_float)), int(seed) & 0xFFFFFFFF, int(drone) & 0xFFFFFFFF,
                      out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def split_half(state: np.ndarray, seed: int = 0, drone_id_offset: int = 0):
    """[14, ld] fp32 SoA -> (pos [3, ld] fp32, sh [11 * ld] uint16): the storage of the fp16 kernels (five pair rows, then the
    thrust row) as fpv_pack_half writes it with rounding seed `seed` and global ids drone_id_offset + i - what
    fpv_reset_kernel leaves behind for rounding_seed = seed (the reset attitude (1, 0, 0, 0) and a zero velocity are exact
    in the format; a general reset pose is rounded stochastically, hence the seed)."""
    L = lib()
    L.fpvl_pack_rows.argtypes = [C.POINTER(C.c_float), C.c_int64, C.c_int64, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.c_void_p]
    st = np.ascontiguousarray(state, dtype=np.float32)
    ld = st.shape[1]
    pos = np.zeros((3, ld), dtype=np.float32)
    sh = np.zeros(abi.FPV_HALF_HALVES * ld, dtype=np.uint16)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    L.fpvl_pack_rows(fp(st), ld, ld, int(seed) & 0xFFFFFFFF, int(drone_id_offset) & 0xFFFFFFFF, fp(pos), sh.ctypes.data)
    return pos, sh


def join_half(pos: np.ndarray, sh: np.ndarray) -> np.ndarray:
    """(pos [3, ld], sh [11 * ld] uint16) -> [14, ld] fp32: the state as the fp16 kernels read it (fpv_unpack_half: v with
    its low words, q rebuilt from its three stored components)."""
    L = lib()
    L.fpvl_unpack_rows.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_float)]
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    ld = pos.shape[1]
    sh = np.ascontiguousarray(np.asarray(sh).reshape(-1).view(np.uint16))
    assert sh.size == abi.FPV_HALF_HALVES * ld
    st = np.zeros((14, ld), dtype=np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    L.fpvl_unpack_rows(fp(pos), sh.ctypes.data, ld, ld, fp(st))
    return st


def stick_noise(p, n: int, steps: int, noise_seed: int = 0, drone_id_offset: int = 0, base_actions=None,
                step0: int = 0, ns: Optional[np.ndarray] = None):
    """Host build of the in-kernel generator.  Returns (applied [steps, n, 4] fp32, ns [4, ld] fp32)."""
    L = lib()
    L.fpvl_stick_noise.argtypes = [C.POINTER(abi.FpvParams), C.c_int64, C.c_int, C.POINTER(C.c_float), C.c_int64,
                                   C.c_void_p, C.POINTER(C.c_float), C.c_uint64]
    L.fpvl_stick_noise.restype = C.c_int
    ld = (n + 63) // 64 * 64 if ns is None else ns.shape[1]
    ns = np.zeros((4, ld), dtype=np.float32) if ns is None else ns
    applied = np.zeros((steps, n, 4), dtype=np.float32)
    base = None if base_actions is None else np.ascontiguousarray(base_actions, dtype=np.float32)
    cp = abi.pack_params(p, stick_noise=True, noise_seed=noise_seed, drone_id_offset=drone_id_offset)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    rc = L.fpvl_stick_noise(C.byref(cp), n, steps, fp(ns), ld, None if base is None else base.ctypes.data, fp(applied),
                            int(step0) & (2 ** 64 - 1))
    if rc != 0:
        raise RuntimeError(f"fpvl_stick_noise failed with {rc}")
    return applied, ns


def sincos_reduced(x: np.ndarray):
    """(sin x, cos x) in the kernel's own fp32 arithmetic for angles of any size (fpv_sincos_reduced)."""
    L = lib()
    L.fpvl_sincos_reduced.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    x = np.asarray(x, dtype=np.float32)
    s, c = np.zeros_like(x), np.zeros_like(x)
    a, b = C.c_float(), C.c_float()
    for i, v in enumerate(x):
        L.fpvl_sincos_reduced(float(v), C.byref(a), C.byref(b))
        s[i], c[i] = a.value, b.value
    return s, c


def round_seed(base: int, step: int) -> int:
    L = lib()
    L.fpvl_round_seed.argtypes = [C.c_uint32, C.c_uint64]
    L.fpvl_round_seed.restype = C.c_uint32
    return int(L.fpvl_round_seed(int(base) & 0xFFFFFFFF, int(step) & (2 ** 64 - 1)))


def philox(ctr, key, rounds: int = 10):
    """Philox4x32 with 10 (the Random123 reference function) or 7 rounds (what the noise generator runs)."""
    L = lib()
    out = (C.c_uint32 * 4)()
    L.fpvl_philox((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out, int(rounds))
    return [int(x) for x in out]


def noise_philox_rounds() -> int:
    return int(lib().fpvl_noise_philox_rounds())


def normal_from_words(w: np.ndarray) -> np.ndarray:
    """The generator's table-driven inverse normal CDF (fpv_normal_from_word) on an array of 32-bit words."""
    L = lib()
    L.fpvl_normal_from_words.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_int64]
    w = np.ascontiguousarray(w, dtype=np.uint32)
    z = np.zeros(w.shape, dtype=np.float32)
    L.fpvl_normal_from_words(w.ctypes.data_as(C.POINTER(C.c_uint32)), z.ctypes.data_as(C.POINTER(C.c_float)), w.size)
    return z


def pid_run(gains, current: np.ndarray, target: np.ndarray):
    """components.PID in the kernel's fp32 arithmetic (fpv_pid_axis<float, 1>) over a sequence.
    Returns (out [T] fp32, state [4] fp32 = integral, prev_derivative, previous_error, is_first)."""
    L = lib()
    L.fpvl_pid_run.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float),
                               C.POINTER(C.c_float), C.POINTER(C.c_float)]
    k = np.ascontiguousarray(gains, dtype=np.float64)
    cur = np.ascontiguousarray(current, dtype=np.float32)
    tgt = np.ascontiguousarray(target, dtype=np.float32)
    st = np.array([0, 0, 0, 1], dtype=np.float32)
    out = np.zeros(len(cur), dtype=np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    L.fpvl_pid_run(k.ctypes.data_as(C.POINTER(C.c_double)), fp(st), len(cur), fp(cur), fp(tgt), fp(out))
    return out, st


def quat_from_rot(R) -> np.ndarray:
    """[...,3,3] rotation matrices -> [...,4] (w,x,y,z) in the kernel's fp32 arithmetic (fpv_quat_from_rot)."""
    L = lib()
    L.fpvl_quat_from_rot.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    R = np.ascontiguousarray(R, dtype=np.float32).reshape(-1, 9)
    q = np.zeros((R.shape[0], 4), dtype=np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    for i in range(R.shape[0]):
        L.fpvl_quat_from_rot(fp(R[i]), fp(q[i]))
    return q
